/* frontend.c -- the block that replaces dumphfdl's fft block AND all of its channel threads (src/fft.c, src/hfdl.c):
 * it drains the input ring exactly like fft_thread (src/fft.c:38-54), hands each block of input_size samples to the
 * GPU front end (include/hfdl_gpu.h) and turns the PDUs that come back into pdu_decoder_queue_push() calls the way
 * dispatch_pdu does (src/hfdl.c:1058-1080). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hfdl_host.h"
#include "hfdl_gpu.h"
#include "host_internal.h"

struct gpu_fft_block {
	struct block block;
	int32_t decimation;
	float transition_bw;
	int device;
	hfdl_gpu_geometry geo;
};

static int g_device = 0;
void hfdl_frontend_set_device(int device) { g_device = device; }

/* ---- libcsdr helpers main() needs (src/libcsdr.c:135-144) ---- */

int32_t compute_fft_decimation_rate(int32_t sample_rate, int32_t target_rate)
{
	int32_t whole = (int32_t)floorf((float)sample_rate / (float)target_rate);
	for (int i = 0; i < 31; i++) if (whole < (1 << i)) return (1 << i) / 2;
	return -1;
}

float compute_filter_relative_transition_bw(int32_t sample_rate, int32_t transition_bw_hz)
{
	return (float)transition_bw_hz / (float)sample_rate;
}

/* ---- channel slots ---- */

#define MAX_SLOTS 4096
static struct hfdl_channel_slot *g_slots[MAX_SLOTS];
static size_t g_slot_cnt;
static pthread_mutex_t g_slot_lock = PTHREAD_MUTEX_INITIALIZER;

void hfdl_init_globals(void) { /* preamble sequences and filter tables live in libhfdl_gpu.so (built at create time) */ }

static void *channel_idle_thread(void *ctx)
{
	struct block *block = ctx;
	struct shared_buffer *in = &block->consumer.in->shared_buffer;
	pthread_barrier_wait(in->consumers_ready);      /* "all consumers initialised", src/hfdl.c:663 / src/fft.c:36 */
	pthread_barrier_wait(in->data_ready);           /* released once, by block_connection_one2many_shutdown() */
	block->running = false;
	return NULL;
}

struct block *hfdl_channel_create(int32_t sample_rate, int32_t pre_decimation_rate, float transition_bw,
		int32_t centerfreq, int32_t frequency)
{
	if (sample_rate <= 0 || pre_decimation_rate <= 0) return NULL;
	struct hfdl_channel_slot *s = hfdl_xcalloc(1, sizeof(*s));
	s->sample_rate = sample_rate; s->pre_decimation_rate = pre_decimation_rate; s->transition_bw = transition_bw;
	s->centerfreq = centerfreq; s->frequency = frequency;
	s->block.producer.type = PRODUCER_NONE;
	s->block.consumer.type = CONSUMER_MULTI;
	s->block.consumer.min_ru = 0;
	s->block.thread_routine = channel_idle_thread;
	pthread_mutex_lock(&g_slot_lock);
	if (g_slot_cnt == MAX_SLOTS) { pthread_mutex_unlock(&g_slot_lock); free(s); return NULL; }
	g_slots[g_slot_cnt++] = s;
	pthread_mutex_unlock(&g_slot_lock);
	return &s->block;
}

void hfdl_channel_destroy(struct block *channel_block)
{
	if (channel_block == NULL) return;
	struct hfdl_channel_slot *s = container_of(channel_block, struct hfdl_channel_slot, block);
	pthread_mutex_lock(&g_slot_lock);
	for (size_t i = 0; i < g_slot_cnt; i++) if (g_slots[i] == s) { g_slots[i] = g_slots[--g_slot_cnt]; break; }
	pthread_mutex_unlock(&g_slot_lock);
	free(s);
}

size_t hfdl_channels_on_connection(struct block_connection *conn, struct hfdl_channel_slot **out, size_t max)
{
	size_t n = 0;
	pthread_mutex_lock(&g_slot_lock);
	for (size_t i = 0; i < g_slot_cnt && n < max; i++) if (g_slots[i]->block.consumer.in == conn) out[n++] = g_slots[i];
	pthread_mutex_unlock(&g_slot_lock);
	return n;
}

void hfdl_print_summary(void) {}
int32_t hfdl_nf_stats_thread_start(struct block **channel_block_list, int32_t channel_cnt)
{
	(void)channel_block_list; (void)channel_cnt;
	return 0;            /* StatsD gauges are out of scope */
}

/* ---- the front-end thread ---- */

static void push_pdu(const hfdl_gpu_pdu *p, const struct timeval *t0)
{
	struct metadata *m = hfdl_pdu_metadata_create();
	struct hfdl_pdu_metadata *hm = container_of(m, struct hfdl_pdu_metadata, metadata);
	hm->version = 1;
	hm->freq = p->freq;
	hm->freq_err_hz = p->freq_err_hz;
	hm->rssi = p->rssi_db;
	hm->noise_floor = p->noise_floor_db;
	hm->bit_rate = p->bit_rate;
	hm->slot = p->slot;
	/* start of frame = A2 detection - (prekey + 2 A) symbols (src/hfdl.c:657-660), on the stream's sample clock */
	double t = (double)t0->tv_sec + 1e-6 * (double)t0->tv_usec + (double)p->sample_index / (HFDL_SYMBOL_RATE * SPS)
		- (448.0 + 2 * 127.0) / HFDL_SYMBOL_RATE;
	m->rx_timestamp.tv_sec = (time_t)floor(t);
	m->rx_timestamp.tv_usec = (suseconds_t)((t - floor(t)) * 1e6);
	uint8_t *copy = hfdl_xcalloc((size_t)p->len ? (size_t)p->len : 1, 1);
	memcpy(copy, p->octets, (size_t)p->len);
	pdu_decoder_queue_push(m, octet_string_new(copy, (size_t)p->len), 0);
}

static void *frontend_thread(void *ctx)
{
	struct block *block = ctx;
	struct gpu_fft_block *fb = container_of(block, struct gpu_fft_block, block);
	struct circ_buffer *ring = &block->consumer.in->circ_buffer;
	struct block_connection *down = block->producer.out;
	hfdl_gpu_frontend *fe = NULL;
	float complex *stage = NULL;
	hfdl_gpu_pdu *pdus = NULL;
	const int32_t max_pdus = 1024;

	struct hfdl_channel_slot *slots[MAX_SLOTS];
	size_t nch = hfdl_channels_on_connection(down, slots, MAX_SLOTS);
	int32_t *freqs = hfdl_xcalloc(nch ? nch : 1, sizeof(int32_t));
	for (size_t i = 0; i < nch; i++) freqs[i] = slots[i]->frequency;
	int ok = nch > 0 && hfdl_gpu_frontend_create(&fe, fb->device, slots[0]->sample_rate, slots[0]->centerfreq, freqs, (int32_t)nch) == 0;
	if (!ok) {
		fprintf(stderr, "GPU front end: %s\n", nch ? hfdl_gpu_last_error() : "no channels connected");
		do_exit = 1;
	} else {
		hfdl_gpu_frontend_geometry(fe, &fb->geo);
		hfdl_gpu_frontend_enable_taps(fe, 0);
		if (hfdl_gpu_host_alloc((void **)&stage, sizeof(float complex) * (size_t)fb->geo.input_size) != 0)
			stage = hfdl_xcalloc((size_t)fb->geo.input_size, sizeof(float complex));
		pdus = hfdl_xcalloc((size_t)max_pdus, sizeof(*pdus));
	}
	pthread_barrier_wait(down->shared_buffer.consumers_ready);
	struct timeval t0;
	gettimeofday(&t0, NULL);
	const size_t need = ok ? (size_t)fb->geo.input_size : 1;
	for (;;) {
		pthread_mutex_lock(ring->mutex);
		/* shutdown is honoured only when there is not a whole block left, so buffered samples are flushed (src/fft.c:39-48) */
		while (hfdl_ring_size(ring->buf) < need) {
			if (block_connection_is_shutdown_signaled(block->consumer.in)) { pthread_mutex_unlock(ring->mutex); goto shutdown; }
			pthread_cond_wait(ring->cond, ring->mutex);
		}
		if (ok) hfdl_ring_read(ring->buf, stage, need); else hfdl_ring_read(ring->buf, (float complex[1]){0}, 1);
		pthread_mutex_unlock(ring->mutex);
		if (!ok) continue;
		if (hfdl_gpu_frontend_push_block(fe, (const float *)stage, need, 0) != 0) {
			fprintf(stderr, "GPU front end: %s\n", hfdl_gpu_last_error());
			do_exit = 1;
			ok = 0;
			continue;
		}
		int32_t n = 0;
		if (hfdl_gpu_frontend_poll_pdus(fe, pdus, max_pdus, &n) == 0)
			for (int32_t i = 0; i < n; i++) push_pdu(&pdus[i], &t0);
	}
shutdown:
	block_connection_one2many_shutdown(down);
	if (fe) hfdl_gpu_frontend_destroy(fe);
	if (stage) hfdl_gpu_host_free(stage);
	free(pdus);
	free(freqs);
	block->running = false;
	return NULL;
}

struct block *fft_create(int32_t decimation, float transition_bw)
{
	struct gpu_fft_block *fb = hfdl_xcalloc(1, sizeof(*fb));
	if (hfdl_gpu_plan_geometry(decimation, transition_bw, &fb->geo) != 0) {
		fprintf(stderr, "Error in fastddc_init()");
		free(fb);
		return NULL;
	}
	fb->decimation = decimation;
	fb->transition_bw = transition_bw;
	fb->device = g_device;
	fb->block.producer.type = PRODUCER_MULTI;
	fb->block.producer.max_tu = (size_t)fb->geo.fft_size;
	fb->block.consumer.type = CONSUMER_SINGLE;
	fb->block.consumer.min_ru = (size_t)fb->geo.fft_size;
	fb->block.thread_routine = frontend_thread;
	return &fb->block;
}

void fft_destroy(struct block *fft_block)
{
	if (fft_block) free(container_of(fft_block, struct gpu_fft_block, block));
}
