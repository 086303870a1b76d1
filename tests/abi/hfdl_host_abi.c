/* Layout and enumerator pins of include/hfdl_host.h (LP64 / x86-64, the platform dumphfdl and this library are built for).
 * Every expected number is derived by hand from the REFERENCE declaration named beside it -- the declarations in
 * the .h files under /root/reference/src, field by field, natural alignment -- so a host program compiled against dumphfdl's own headers and
 * libhfdl_host.so compiled against include/hfdl_host.h agree on every struct they pass to each other.  Compiled twice by
 * tests/test_host_abi_cpu.py: as is (the reference's default build) and with -DWITH_SOAPYSDR.  Syntax check only; nothing runs. */
#include <stddef.h>
#include "hfdl_host.h"

#define SIZE(T, N) _Static_assert(sizeof(T) == (N), "sizeof(" #T ") != " #N)
#define OFF(T, F, N) _Static_assert(offsetof(T, F) == (N), "offsetof(" #T ", " #F ") != " #N)
#define VAL(E, N) _Static_assert((int)(E) == (N), #E " != " #N)

/* src/block.h:13-25  enum producer_type / consumer_type: NONE = 0, SINGLE, MULTI, MAX */
VAL(PRODUCER_NONE, 0); VAL(PRODUCER_SINGLE, 1); VAL(PRODUCER_MULTI, 2); VAL(PRODUCER_MAX, 3);
VAL(CONSUMER_NONE, 0); VAL(CONSUMER_SINGLE, 1); VAL(CONSUMER_MULTI, 2); VAL(CONSUMER_MAX, 3);

/* src/block.h:27-31  struct circ_buffer { cbuffercf buf (liquid: a pointer typedef); pthread_cond_t *cond; pthread_mutex_t *mutex; } */
SIZE(struct circ_buffer, 24); OFF(struct circ_buffer, buf, 0); OFF(struct circ_buffer, cond, 8); OFF(struct circ_buffer, mutex, 16);
/* src/block.h:33-37  struct shared_buffer { float complex *buf; pthread_barrier_t *data_ready, *consumers_ready; } */
SIZE(struct shared_buffer, 24); OFF(struct shared_buffer, buf, 0); OFF(struct shared_buffer, data_ready, 8); OFF(struct shared_buffer, consumers_ready, 16);
/* src/block.h:39-45  struct block_connection { union { circ_buffer; shared_buffer; } (24); uint32_t flags; } -> 28, padded to 32 */
SIZE(struct block_connection, 32); OFF(struct block_connection, circ_buffer, 0); OFF(struct block_connection, shared_buffer, 0); OFF(struct block_connection, flags, 24);
/* src/block.h:48  #define BLOCK_CONNECTION_SHUTDOWN (1 << 0) */
VAL(BLOCK_CONNECTION_SHUTDOWN, 1);
/* src/block.h:50-54  struct producer { struct block_connection *out; size_t max_tu; enum producer_type type; } -> 20, padded to 24 */
SIZE(struct producer, 24); OFF(struct producer, out, 0); OFF(struct producer, max_tu, 8); OFF(struct producer, type, 16);
/* src/block.h:56-60  struct consumer { struct block_connection *in; size_t min_ru; enum consumer_type type; } */
SIZE(struct consumer, 24); OFF(struct consumer, in, 0); OFF(struct consumer, min_ru, 8); OFF(struct consumer, type, 16);
/* src/block.h:62-68  struct block { consumer (24); producer (24); pthread_t thread (unsigned long); void *(*thread_routine)(void *); bool running; } */
SIZE(struct block, 72); OFF(struct block, consumer, 0); OFF(struct block, producer, 24); OFF(struct block, thread, 48); OFF(struct block, thread_routine, 56); OFF(struct block, running, 64);

/* src/input-common.h:8-15  typedef enum { INPUT_TYPE_UNDEF, [WITH_SOAPYSDR: INPUT_TYPE_SOAPYSDR,] INPUT_TYPE_FILE, INPUT_TYPE_MAX } input_type */
VAL(INPUT_TYPE_UNDEF, 0);
#ifdef WITH_SOAPYSDR
VAL(INPUT_TYPE_SOAPYSDR, 1); VAL(INPUT_TYPE_FILE, 2); VAL(INPUT_TYPE_MAX, 3);
#else
VAL(INPUT_TYPE_FILE, 1); VAL(INPUT_TYPE_MAX, 2);
#endif
/* src/input-common.h:17-23  typedef enum { SFMT_UNDEF = 0, SFMT_CU8, SFMT_CS16, SFMT_CF32, SFMT_MAX } sample_format */
VAL(SFMT_UNDEF, 0); VAL(SFMT_CU8, 1); VAL(SFMT_CS16, 2); VAL(SFMT_CF32, 3); VAL(SFMT_MAX, 4);
/* src/input-common.h:27-40  struct input_cfg { char *source, *gain_elements, *antenna, *device_settings; double gain, correction;
 *                            int32_t sample_rate, centerfreq, freq_offset, read_buffer_size; input_type type; sample_format sfmt; } */
SIZE(struct input_cfg, 72);
OFF(struct input_cfg, source, 0); OFF(struct input_cfg, gain_elements, 8); OFF(struct input_cfg, antenna, 16); OFF(struct input_cfg, device_settings, 24);
OFF(struct input_cfg, gain, 32); OFF(struct input_cfg, correction, 40);
OFF(struct input_cfg, sample_rate, 48); OFF(struct input_cfg, centerfreq, 52); OFF(struct input_cfg, freq_offset, 56); OFF(struct input_cfg, read_buffer_size, 60);
OFF(struct input_cfg, type, 64); OFF(struct input_cfg, sfmt, 68);
/* src/input-common.h:44-49  struct input_vtable { create; init; destroy; rx_thread_routine; } four function pointers */
SIZE(struct input_vtable, 32); OFF(struct input_vtable, create, 0); OFF(struct input_vtable, init, 8); OFF(struct input_vtable, destroy, 16); OFF(struct input_vtable, rx_thread_routine, 24);
/* src/input-common.h:53-61  struct input { struct block block (72); struct input_vtable *vtable; struct input_cfg *config;
 *                            convert_sample_buffer_fun convert_sample_buffer; size_t overflow_count; float full_scale; int32_t bytes_per_sample; } */
SIZE(struct input, 112);
OFF(struct input, block, 0); OFF(struct input, vtable, 72); OFF(struct input, config, 80); OFF(struct input, convert_sample_buffer, 88);
OFF(struct input, overflow_count, 96); OFF(struct input, full_scale, 104); OFF(struct input, bytes_per_sample, 108);

/* src/metadata.h:5-8  struct metadata { struct metadata_vtable *vtable; struct timeval rx_timestamp (two longs); } */
SIZE(struct metadata, 24); OFF(struct metadata, vtable, 0); OFF(struct metadata, rx_timestamp, 8);
/* src/metadata.h:10-13  struct metadata_vtable { copy; destroy; } */
SIZE(struct metadata_vtable, 16); OFF(struct metadata_vtable, copy, 0); OFF(struct metadata_vtable, destroy, 8);
/* src/pdu.h:8-17  struct hfdl_pdu_metadata { struct metadata metadata (24); int32_t version, freq, bit_rate; float freq_err_hz, rssi,
 *                  noise_floor; char slot; } -> 49, padded to 56 */
SIZE(struct hfdl_pdu_metadata, 56);
OFF(struct hfdl_pdu_metadata, metadata, 0); OFF(struct hfdl_pdu_metadata, version, 24); OFF(struct hfdl_pdu_metadata, freq, 28); OFF(struct hfdl_pdu_metadata, bit_rate, 32);
OFF(struct hfdl_pdu_metadata, freq_err_hz, 36); OFF(struct hfdl_pdu_metadata, rssi, 40); OFF(struct hfdl_pdu_metadata, noise_floor, 44); OFF(struct hfdl_pdu_metadata, slot, 48);
/* src/util.h:119-122  struct octet_string { uint8_t *buf; size_t len; } */
SIZE(struct octet_string, 16); OFF(struct octet_string, buf, 0); OFF(struct octet_string, len, 8);

/* src/hfdl.h:6-8  SPS 3, HFDL_SYMBOL_RATE 1800, HFDL_CHANNEL_TRANSITION_BW_HZ 250 */
VAL(SPS, 3); VAL(HFDL_SYMBOL_RATE, 1800); VAL(HFDL_CHANNEL_TRANSITION_BW_HZ, 250);

/* the signatures a host program calls, as the reference declares them (src/block.h:70-81, src/input-common.h:63-67, src/fft.h:31-32,
 * src/hfdl.h:10-15, src/pdu.h:38-39): assigning to a pointer of the reference's type fails to compile on any mismatch */
static int32_t (*const p_connect)(struct block *, struct block *) = block_connect_one2one;
static int32_t (*const p_connect_many)(struct block *, size_t, struct block *[]) = block_connect_one2many;
static int32_t (*const p_start)(struct block *) = block_start;
static bool (*const p_any)(size_t, struct block *[]) = block_set_is_any_running;
static struct block *(*const p_input_create)(struct input_cfg *) = input_create;
static int32_t (*const p_input_init)(struct block *) = input_init;
static struct block *(*const p_fft_create)(int32_t, float) = fft_create;
static struct block *(*const p_chan_create)(int32_t, int32_t, float, int32_t, int32_t) = hfdl_channel_create;
static void (*const p_push)(struct metadata *, struct octet_string *, uint32_t) = pdu_decoder_queue_push;
static void (*const p_produce)(struct circ_buffer *, float complex *, size_t) = complex_samples_produce;
const void *hfdl_host_abi_uses[] = { (const void *)p_connect, (const void *)p_connect_many, (const void *)p_start, (const void *)p_any, (const void *)p_input_create,
	(const void *)p_input_init, (const void *)p_fft_create, (const void *)p_chan_create, (const void *)p_push, (const void *)p_produce };
