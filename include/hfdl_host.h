/*
 * hfdl_host.h -- the host-side (plain C11) API of the MI355X HFDL front end: libhfdl_host.so
 *
 * It keeps the reference's block / input / channel / PDU hand-off interface so that dumphfdl's main.c wiring
 * (src/main.c:687-774) and everything downstream of pdu_decoder_queue_push() (parsers, formatters, outputs) read the
 * same, while the data path between complex_samples_produce() and pdu_decoder_queue_push() runs on the GPU through
 * include/hfdl_gpu.h.  Names, argument meaning, return values and ownership rules are the reference's:
 *
 *   struct block / producer / consumer / block_connection      src/block.h:13-68
 *   block_connect_one2one / one2many, block_start, ...         src/block.h:70-81, src/block.c:55-193
 *   struct input_cfg / input / input_vtable, input_create ...  src/input-common.h:8-67, src/input-common.c
 *   complex_samples_produce, sample converters                 src/input-helpers.h:9-14, src/input-helpers.c:10-156
 *   file input                                                 src/input-file.c:15-119
 *   fft_create / fft_destroy                                   src/fft.h:31-32, src/fft.c:70-95
 *   hfdl_init_globals / hfdl_channel_create / _destroy ...     src/hfdl.h:10-15
 *   struct metadata, struct hfdl_pdu_metadata                  src/metadata.h:5-13, src/pdu.h:8-17
 *   struct octet_string, octet_string_new                      src/util.h:119-123, src/util.c:100-105
 *   pdu_decoder_queue_push (CALLED by this library, PROVIDED by the host program) src/pdu.h:39, src/pdu.c:37-43
 *
 * Differences, all behind the same signatures:
 *   - the ring behind struct circ_buffer is our own cf32 ring (liquid's cbuffercf is not needed);
 *   - fft_create() returns the block that owns the GPU front end; the blocks returned by hfdl_channel_create() only
 *     register a frequency with it -- their threads idle until shutdown, so block_set_is_any_running() keeps working;
 *   - rx_timestamp is derived from the sample clock (the reference uses gettimeofday(), src/hfdl.c:808-809).
 */
#ifndef HFDL_HOST_H
#define HFDL_HOST_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <complex.h>
#include <pthread.h>
#include <sys/time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ block runtime (src/block.h) */

enum producer_type { PRODUCER_NONE = 0, PRODUCER_SINGLE, PRODUCER_MULTI, PRODUCER_MAX };
enum consumer_type { CONSUMER_NONE = 0, CONSUMER_SINGLE, CONSUMER_MULTI, CONSUMER_MAX };

struct hfdl_ring;                       /* cf32 FIFO: the role liquid's cbuffercf plays in the reference */

struct circ_buffer {
	struct hfdl_ring *buf;
	pthread_cond_t *cond;
	pthread_mutex_t *mutex;
};

struct shared_buffer {
	float complex *buf;
	pthread_barrier_t *data_ready;
	pthread_barrier_t *consumers_ready;
};

struct block_connection {
	union {
		struct circ_buffer circ_buffer;
		struct shared_buffer shared_buffer;
	};
	uint32_t flags;
};
#define BLOCK_CONNECTION_SHUTDOWN (1 << 0)

struct producer { struct block_connection *out; size_t max_tu; enum producer_type type; };
struct consumer { struct block_connection *in; size_t min_ru; enum consumer_type type; };

struct block {
	struct consumer consumer;
	struct producer producer;
	pthread_t thread;
	void *(*thread_routine)(void *);
	bool running;
};

int32_t block_connect_one2one(struct block *source, struct block *sink);                      /* 1 on success, 0 on failure */
int32_t block_connect_one2many(struct block *source, size_t sink_count, struct block *sinks[]); /* number of sinks connected */
void    block_disconnect_one2one(struct block *source, struct block *sink);
void    block_disconnect_one2many(struct block *source, size_t sink_count, struct block *sinks[]);
int32_t block_start(struct block *block);                                                     /* 1 if the thread started */
int32_t block_set_start(size_t block_cnt, struct block *block[]);
void    block_connection_one2one_shutdown(struct block_connection *connection);
void    block_connection_one2many_shutdown(struct block_connection *connection);
bool    block_connection_is_shutdown_signaled(struct block_connection *connection);
bool    block_is_running(struct block *block);
bool    block_set_is_any_running(size_t block_cnt, struct block *blocks[]);

/* the ring itself (exposed for tests and for other producers) */
struct hfdl_ring *hfdl_ring_create(size_t capacity);
void   hfdl_ring_destroy(struct hfdl_ring *r);
size_t hfdl_ring_size(const struct hfdl_ring *r);
size_t hfdl_ring_space_available(const struct hfdl_ring *r);
size_t hfdl_ring_write(struct hfdl_ring *r, const float complex *src, size_t n);   /* returns samples written */
size_t hfdl_ring_read(struct hfdl_ring *r, float complex *dst, size_t n);          /* returns samples read */

/* ------------------------------------------------------------------ inputs (src/input-common.h, input-helpers.h) */

/* The reference's enum, conditional member included (src/input-common.h:8-15): INPUT_TYPE_FILE is 1 in a default build of the
 * host program and 2 when it is built WITH_SOAPYSDR.  This library carries no SoapySDR code: in a WITH_SOAPYSDR build the slot is
 * filled by the host program with input_vtable_register(INPUT_TYPE_SOAPYSDR, &soapysdr_input_vtable). */
typedef enum {
	INPUT_TYPE_UNDEF,
#ifdef WITH_SOAPYSDR
	INPUT_TYPE_SOAPYSDR,
#endif
	INPUT_TYPE_FILE,
	INPUT_TYPE_MAX
} input_type;
/* libhfdl_host.so is ONE binary for both builds of the host program.  The two entry points that interpret an input_type value bind,
 * THROUGH THE TWO #defines BELOW, to the symbol that reads the caller's numbering.  The rebinding lives in this header only: a
 * WITH_SOAPYSDR host program must compile its callers of input_create() / input_vtable_register() against THIS header (in place of
 * src/input-common.h) -- built against dumphfdl's own header it would call the plain symbols with INPUT_TYPE_FILE = 2, which the
 * plain symbols read in the default numbering (no such type: NULL).  A default build (no SoapySDR) agrees either way. */
#ifdef WITH_SOAPYSDR
#define input_create          input_create_with_soapysdr
#define input_vtable_register input_vtable_register_with_soapysdr
#endif
typedef enum { SFMT_UNDEF = 0, SFMT_CU8, SFMT_CS16, SFMT_CF32, SFMT_MAX } sample_format;

struct input_cfg {
	char *source;
	char *gain_elements, *antenna, *device_settings;
	double gain, correction;
	int32_t sample_rate, centerfreq, freq_offset, read_buffer_size;
	input_type type;
	sample_format sfmt;
};

struct input;
struct input_vtable {
	struct input *(*create)(struct input_cfg *);
	int32_t (*init)(struct input *);
	void (*destroy)(struct input *);
	void *(*rx_thread_routine)(void *);
};
typedef void (*convert_sample_buffer_fun)(struct input *, void *, size_t, float complex *);

struct input {
	struct block block;
	struct input_vtable *vtable;
	struct input_cfg *config;
	convert_sample_buffer_fun convert_sample_buffer;
	size_t overflow_count;
	float full_scale;
	int32_t bytes_per_sample;
};

/* Plug an input implementation into input_create()'s table (input_vtables[], src/input-common.c:12-18): dumphfdl's
 * soapysdr_input_vtable registers under INPUT_TYPE_SOAPYSDR unchanged -- its rx thread only needs complex_samples_produce()
 * and the block fields.  Returns 0, or -1 for a type outside the caller's enum / a NULL or incomplete table.  Not in the reference
 * (which fills the table at compile time). */
int32_t input_vtable_register(input_type type, struct input_vtable const *vtable);

struct input_cfg *input_cfg_create(void);
void          input_cfg_destroy(struct input_cfg *cfg);
struct block *input_create(struct input_cfg *cfg);          /* NULL on error */
int32_t       input_init(struct block *block);              /* 0 ok, negative on error */
void          input_destroy(struct block *block);

size_t  get_sample_size(sample_format format);
float   get_sample_full_scale_value(sample_format format);
convert_sample_buffer_fun get_sample_converter(sample_format format);
sample_format sample_format_from_string(char const *str);
/* copies into the ring under the mutex, drops what does not fit (with a message), signals the consumer */
void    complex_samples_produce(struct circ_buffer *circ_buffer, float complex *samples, size_t num_samples);

/* ------------------------------------------------------------------ channelizer + channels (src/fft.h, src/hfdl.h) */

#define SPS 3
#define HFDL_SYMBOL_RATE 1800
#define HFDL_CHANNEL_TRANSITION_BW_HZ 250

int32_t compute_fft_decimation_rate(int32_t sample_rate, int32_t target_rate);               /* src/libcsdr.c:140-144 */
float   compute_filter_relative_transition_bw(int32_t sample_rate, int32_t transition_bw_hz); /* src/libcsdr.c:135-138 */

struct block *fft_create(int32_t decimation, float transition_bw);
void          fft_destroy(struct block *fft_block);

void          hfdl_init_globals(void);
struct block *hfdl_channel_create(int32_t sample_rate, int32_t pre_decimation_rate, float transition_bw,
		int32_t centerfreq, int32_t frequency);
void          hfdl_channel_destroy(struct block *channel_block);
void          hfdl_print_summary(void);
int32_t       hfdl_nf_stats_thread_start(struct block **channel_block_list, int32_t channel_cnt);

/* seconds between noise-floor gauges (Config.nf_stats_interval, src/main.c:595, src/hfdl.c:1090); 0 = off.  Set before
 * hfdl_nf_stats_thread_start(); not in the reference, which reads its global Config. */
void          hfdl_nf_stats_set_interval(int32_t seconds);

/* PROVIDED BY THE HOST PROGRAM when it is built WITH_STATSD (dumphfdl's src/statsd.c, prototypes src/statsd.h:10-20).
 * The front-end thread calls them with the reference's metric names: "demod.preamble.A2_found", "demod.preamble.M1_found",
 * "demod.preamble.errors.M1_not_found" once per event (src/hfdl.c:818-840) and the "noise_floor" gauge in tenths of
 * -dBFS (src/hfdl.c:1093-1101).  libhfdl_host.so carries weak defaults that do nothing. */
void statsd_counter_per_channel_increment(int32_t freq, char *counter);
void statsd_gauge_per_channel_set(int32_t freq, char *gauge, size_t value);

/* GPU selection for the front end created by the next fft_create() (default 0); not in the reference */
void          hfdl_frontend_set_device(int device);

/* What the front-end thread did, readable once its block has stopped running (not in the reference; hfdl_replay --bench):
 * seconds runs from the first block handed to the GPU to the last PDU handed to pdu_decoder_queue_push(). */
struct hfdl_run_stats {
	uint64_t blocks, samples, pdus;
	double seconds;
	int32_t bytes_per_sample;        /* of the samples as they crossed PCIe: 8 cf32, 4 cs16, 2 cu8 (converted on the device) */
	int32_t channels, block_samples;
	int32_t zero_copy;               /* 1: blocks were DMA'd straight out of the page-locked input ring */
	double wait_input_s, push_s, collect_s, release_s;   /* the front-end thread's time: waiting for the producer, enqueueing,
	                                                        collecting PDUs (waits for the previous block), releasing ring slots (waits for DMA) */
	/* what the device found when it walked the LPDU lists of the MPDUs with a good header FCS (hfdl_gpu_pdu.lpdus_*): informational --
	 * dumphfdl's own lpdu_parse emits the StatsD events for these downstream */
	uint64_t mpdus_walked, lpdus_processed, lpdus_good, lpdus_bad_fcs;
	/* how often the thread took the source for live (next block not there after the grace period) and drained the pipeline, and
	 * the time spent in that grace period */
	uint64_t drains;
	double grace_s;
};
void          hfdl_frontend_run_stats(struct hfdl_run_stats *out);
/* replay a regular input file this many times back to back (default 1); not in the reference */
void          hfdl_file_input_set_loops(int loops);

/* ------------------------------------------------------------------ downstream hand-off (src/pdu.h, metadata.h, util.h) */

struct metadata_vtable;
struct metadata { struct metadata_vtable *vtable; struct timeval rx_timestamp; };
struct metadata_vtable {
	struct metadata *(*copy)(struct metadata const *);
	void (*destroy)(struct metadata *);
};

struct hfdl_pdu_metadata {
	struct metadata metadata;
	int32_t version;
	int32_t freq;
	int32_t bit_rate;
	float freq_err_hz;
	float rssi;
	float noise_floor;
	char slot;
};

struct octet_string { uint8_t *buf; size_t len; };
struct octet_string *octet_string_new(void *buf, size_t len);
void                 octet_string_destroy(struct octet_string *ostring);
struct metadata     *hfdl_pdu_metadata_create(void);

/* PROVIDED BY THE HOST PROGRAM (dumphfdl's src/pdu.c:37-43).  Ownership of both arguments passes to the callee.
 * libhfdl_host.so carries a weak default that prints one line per PDU, so small tools link without the parsers. */
void pdu_decoder_queue_push(struct metadata *metadata, struct octet_string *pdu, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
