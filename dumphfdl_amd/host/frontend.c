/* frontend.c -- the block that replaces dumphfdl's fft block AND all of its channel threads (src/fft.c, src/hfdl.c):
 * it drains the input ring exactly like fft_thread (src/fft.c:38-54), hands each block of input_size samples to the
 * GPU front end (include/hfdl_gpu.h) and turns the PDUs that come back into pdu_decoder_queue_push() calls the way
 * dispatch_pdu does (src/hfdl.c:1058-1080). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdbool.h>
#include <unistd.h>
#include <pthread.h>
#include <time.h>
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

/* ---- StatsD-style observability (src/hfdl.c:818-840, 1082-1105): fed from the device-resident channel state ---- */

__attribute__((weak)) void statsd_counter_per_channel_increment(int32_t freq, char *counter) { (void)freq; (void)counter; }
__attribute__((weak)) void statsd_gauge_per_channel_set(int32_t freq, char *gauge, size_t value) { (void)freq; (void)gauge; (void)value; }

static struct {
	pthread_mutex_t lock;
	hfdl_gpu_channel_stats *last;       /* newest snapshot, one entry per channel of the front end */
	int32_t cnt;
	uint64_t a2, m1, m1_missing, frames; /* totals over all channels */
} g_obs = { .lock = PTHREAD_MUTEX_INITIALIZER };

static int32_t g_nf_interval;
void hfdl_nf_stats_set_interval(int32_t seconds) { g_nf_interval = seconds; }

/* one strided device read per block; emits one increment per new event, like the per-symbol calls of the reference */
static void publish_counters(hfdl_gpu_frontend *fe, hfdl_gpu_channel_stats *now, int32_t nch)
{
	int32_t n = 0;
	if (hfdl_gpu_frontend_all_channel_stats(fe, now, nch, &n) != 0 || n != nch) return;
	pthread_mutex_lock(&g_obs.lock);
	if (g_obs.last == NULL || g_obs.cnt != nch) {
		free(g_obs.last);
		g_obs.last = hfdl_xcalloc((size_t)nch, sizeof(*g_obs.last));
		g_obs.cnt = nch;
	}
	for (int32_t i = 0; i < nch; i++) {
		const hfdl_gpu_channel_stats *o = &g_obs.last[i];
		for (uint32_t k = o->a2_found; k != now[i].a2_found; k++) statsd_counter_per_channel_increment(now[i].freq, "demod.preamble.A2_found");
		for (uint32_t k = o->m1_found; k != now[i].m1_found; k++) statsd_counter_per_channel_increment(now[i].freq, "demod.preamble.M1_found");
		for (uint32_t k = o->m1_not_found; k != now[i].m1_not_found; k++) statsd_counter_per_channel_increment(now[i].freq, "demod.preamble.errors.M1_not_found");
		g_obs.a2 += now[i].a2_found - o->a2_found;
		g_obs.m1 += now[i].m1_found - o->m1_found;
		g_obs.m1_missing += now[i].m1_not_found - o->m1_not_found;
		g_obs.frames += now[i].frames - o->frames;
	}
	memcpy(g_obs.last, now, sizeof(*now) * (size_t)nch);
	pthread_mutex_unlock(&g_obs.lock);
}

/* the reference's debug summary, same lines (src/hfdl.c:563-573): totals over all channels from the newest device snapshot; the
 * correlation averages are weighted by each channel's detection count, as one global S.A1_corr_total / S.A1_found is */
void hfdl_print_summary(void)
{
#ifdef DEBUG
	pthread_mutex_lock(&g_obs.lock);
	uint64_t a1 = 0, a2 = 0, m1 = 0, bad = 0, total = 0;
	double c1 = 0, c2 = 0, cm = 0;
	for (int32_t i = 0; i < g_obs.cnt; i++) {
		const hfdl_gpu_channel_stats *s = &g_obs.last[i];
		a1 += s->a1_found; a2 += s->a2_found; m1 += s->m1_found;
		c1 += (double)s->a1_corr_avg * s->a1_found; c2 += (double)s->a2_corr_avg * s->a2_found; cm += (double)s->m1_corr_avg * s->m1_found;
		bad += s->train_bits_bad; total += s->train_bits_total;
	}
	fprintf(stderr, "A1_found:\t\t%llu\nA2_found:\t\t%llu\nM1_found:\t\t%llu\n", (unsigned long long)a1, (unsigned long long)a2, (unsigned long long)m1);
	fprintf(stderr, "A1_corr_avg:\t\t%4.3f\n", a1 > 0 ? c1 / (double)a1 : 0.0);
	fprintf(stderr, "A2_corr_avg:\t\t%4.3f\n", a2 > 0 ? c2 / (double)a2 : 0.0);
	fprintf(stderr, "M1_corr_avg:\t\t%4.3f\n", m1 > 0 ? cm / (double)m1 : 0.0);
	fprintf(stderr, "train_bits_bad/total:\t%llu/%llu (%f%%)\n", (unsigned long long)bad, (unsigned long long)total,
			(float)bad / (float)total * 100.f);
	pthread_mutex_unlock(&g_obs.lock);
#endif
}

static void *noise_floor_stats_thread(void *ctx)
{
	(void)ctx;
	for (;;) {
		sleep((unsigned)g_nf_interval);
		pthread_mutex_lock(&g_obs.lock);
		for (int32_t i = 0; i < g_obs.cnt; i++) {
			/* tenths of dBFS, positive: a StatsD gauge takes neither floats nor negative values (src/hfdl.c:1093-1101) */
			float nf = g_obs.last[i].noise_floor_db;
			if (nf <= 0.f) statsd_gauge_per_channel_set(g_obs.last[i].freq, "noise_floor", (size_t)fabsf(roundf(nf * 10.f)));
		}
		pthread_mutex_unlock(&g_obs.lock);
	}
	return NULL;
}

int32_t hfdl_nf_stats_thread_start(struct block **channel_block_list, int32_t channel_cnt)
{
	(void)channel_block_list; (void)channel_cnt;     /* every channel of the front end reports; the list is implicit */
	if (g_nf_interval <= 0) return 0;
	pthread_t th;
	pthread_attr_t attr;
	pthread_attr_init(&attr);
	pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);    /* start_thread() detaches, src/util.c:45-63 */
	int ret = pthread_create(&th, &attr, noise_floor_stats_thread, NULL);
	pthread_attr_destroy(&attr);
	return ret == 0 ? 0 : -1;
}

/* ---- the front-end thread ---- */

static uint64_t g_lpdu_tally[4];      /* MPDUs walked, LPDUs processed / good / bad FCS: front-end thread only, copied into the run stats at shutdown */

static void push_pdu(const hfdl_gpu_pdu *p, const struct timeval *t0)
{
	if (p->fcs_status == HFDL_GPU_FCS_GOOD && p->pdu_kind != HFDL_GPU_KIND_SPDU) {
		g_lpdu_tally[0]++; g_lpdu_tally[1] += p->lpdus_processed; g_lpdu_tally[2] += p->lpdus_good; g_lpdu_tally[3] += p->lpdus_bad_fcs;
	}
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

static struct hfdl_run_stats g_run;
static pthread_mutex_t g_run_lock = PTHREAD_MUTEX_INITIALIZER;

void hfdl_frontend_run_stats(struct hfdl_run_stats *out)
{
	pthread_mutex_lock(&g_run_lock);
	*out = g_run;
	pthread_mutex_unlock(&g_run_lock);
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int gpu_format_of(int ring_fmt)
{
	return ring_fmt == SFMT_CS16 ? HFDL_GPU_SFMT_CS16 : ring_fmt == SFMT_CU8 ? HFDL_GPU_SFMT_CU8 : HFDL_GPU_SFMT_CF32;
}

/* give ring slots back to the producer, oldest first, as the DMA engine finishes reading them.  wait_for_one: block until at least
 * the oldest copy is done (the caller is about to sleep and the producer may be waiting for room); otherwise only what is done
 * already.  Returns 0, or -1 if the state of a copy cannot be established (the slot is kept: the producer must never overwrite
 * memory the DMA engine may still read). */
static int release_copied(hfdl_gpu_frontend *fe, struct circ_buffer *ring, size_t need, uint64_t uploads, size_t *leased, bool wait_for_one)
{
	while (*leased > 0) {
		const uint64_t oldest = uploads - *leased;
		if (wait_for_one) {
			if (hfdl_gpu_frontend_input_done_upto(fe, oldest) != 0 && hfdl_gpu_frontend_input_done(fe) != 0) return -1;
			wait_for_one = false;
		} else {
			const int done = hfdl_gpu_frontend_input_copied(fe, oldest);
			if (done < 0) return -1;
			if (done == 0) break;
		}
		pthread_mutex_lock(ring->mutex);
		hfdl_ring_drop(ring->buf, need);
		pthread_mutex_unlock(ring->mutex);
		pthread_cond_signal(ring->cond);
		(*leased)--;
	}
	return 0;
}

/* The front-end thread.  A block never gets copied on the host: the ring in front of this block is page-locked and a whole
 * number of blocks long (block_connect_one2one), so each block is handed to the GPU where it lies -- raw cs16 / cu8 samples
 * included, which the device converts -- and its slot goes back to the producer once the DMA has read it.
 *
 * Uploads run AHEAD of the blocks that compute: whatever whole blocks the ring holds beyond the one being pushed are queued for
 * upload at once (hfdl_gpu_frontend_prefetch_block_raw, up to geometry.prefetch_depth of them), so the copy engine keeps working
 * while this thread waits for the GPU in the collection call -- the fold of a 40 Msps x 256-channel half takes 3 ms, five blocks
 * of PCIe time.
 *
 * Live source or replay?  The pipeline is drained (everything pushed so far folded, demodulated and delivered at once) only when
 * the ring has run EMPTY and the source has stayed silent for a grace period of a quarter of a block's own duration (at most
 * 20 ms), counted from the arrival of the newest block: a live radio delivers a block every block duration and gets its PDUs within that grace; a file reader that hiccups for
 * a millisecond beside a GPU about as fast as itself never drains a filled pipeline (round 4's fixed 0.5 ms grace did, five times
 * in a run, for 19 % of the rate). */
static void *frontend_thread(void *ctx)
{
	struct block *block = ctx;
	struct gpu_fft_block *fb = container_of(block, struct gpu_fft_block, block);
	struct circ_buffer *ring = &block->consumer.in->circ_buffer;
	struct block_connection *down = block->producer.out;
	hfdl_gpu_frontend *fe = NULL;
	void *bounce = NULL;                    /* only for a ring whose blocks are not contiguous (never one made by this library) */
	hfdl_gpu_pdu *pdus = NULL;
	hfdl_gpu_channel_stats *stats = NULL;
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
		pdus = hfdl_xcalloc((size_t)max_pdus, sizeof(*pdus));
		stats = hfdl_xcalloc(nch, sizeof(*stats));
	}
	pthread_barrier_wait(down->shared_buffer.consumers_ready);
	struct timeval t0;
	gettimeofday(&t0, NULL);
	const size_t need = ok ? (size_t)fb->geo.input_size : 1;
	const size_t elem = hfdl_ring_elem_size(ring->buf);
	const int gfmt = gpu_format_of(hfdl_ring_format(ring->buf));
	/* uploads ahead of the pushes: what the library allows, and what the ring can hold beside the block being pushed and room for
	 * the producer to write into */
	size_t depth = 0;
	if (ok) {
		const size_t ring_blocks = hfdl_ring_capacity(ring->buf) / need;
		depth = (size_t)fb->geo.prefetch_depth;
		if (depth > HFDL_GPU_PREFETCH_MAX) depth = HFDL_GPU_PREFETCH_MAX;
		if (ring_blocks < 4) depth = 0; else if (depth > ring_blocks - 3) depth = ring_blocks - 3;
	}
	double grace = ok ? 0.25 * (double)need / (double)slots[0]->sample_rate : 0.0;
	if (grace > 0.020) grace = 0.020;
	if (grace < 0.0005) grace = 0.0005;
	uint64_t k = 0, npdus = 0;
	double t_first = 0, t_last = 0, t_published = 0;
	double s_wait = 0, s_push = 0, s_poll = 0, s_release = 0, s_grace = 0;      /* where this thread's time went (seconds) */
	uint64_t drains = 0;
	size_t leased = 0;                       /* ring slots the DMA engine may still read: the newest `leased` uploads, oldest at the ring's head */
	uint64_t uploads = 0;                    /* host blocks whose copy has been queued, as the GPU library numbers them */
	const void *queued[HFDL_GPU_PREFETCH_MAX + 1];      /* uploads not pushed yet, oldest at q_head */
	size_t q_head = 0, q_len = 0;
	uint64_t undelivered = 0;                /* blocks pushed since the pipeline was last drained */
	struct timespec last_push;               /* when the newest block was taken from the ring and pushed (CLOCK_REALTIME: pthread_cond_timedwait's clock) */
	clock_gettime(CLOCK_REALTIME, &last_push);
	for (;;) {
		const double tw0 = now_s();
		const void *blk = NULL;
		bool from_bounce = false;
		if (q_len > 0) {
			blk = queued[q_head];            /* uploaded ahead: only its kernels remain to be queued */
		} else {
			bool waited_grace = false;
			pthread_mutex_lock(ring->mutex);
			/* shutdown is honoured only when there is not a whole block left, so buffered samples are flushed (src/fft.c:39-48) */
			while (hfdl_ring_size(ring->buf) < (leased + 1) * need) {
				if (block_connection_is_shutdown_signaled(block->consumer.in)) { pthread_mutex_unlock(ring->mutex); goto shutdown; }
				if (ok && leased > 0) {
					/* the producer may be out of room: slots whose upload is done go back before this thread sleeps */
					pthread_mutex_unlock(ring->mutex);
					const double tr = now_s();
					if (release_copied(fe, ring, need, uploads, &leased, true) != 0) {
						fprintf(stderr, "GPU front end: %s\n", hfdl_gpu_last_error());
						do_exit = 1;
						ok = 0;
					}
					s_release += now_s() - tr;
					pthread_mutex_lock(ring->mutex);
					continue;
				}
				if (ok && undelivered > 0 && !waited_grace) {
					/* the ring is empty: a live source, or a reader catching its breath?  Wait the grace period, then deliver */
					/* the grace period counts from the moment the newest block was PUSHED, not from here (the wait for its upload
					 * lies in between): a live block's PDUs leave grace after its arrival, whatever the copy took */
					struct timespec until = last_push;
					until.tv_nsec += (long)(grace * 1e9);
					while (until.tv_nsec >= 1000000000) { until.tv_sec++; until.tv_nsec -= 1000000000; }
					const double tg = now_s();
					int rc = 0;
					while (rc == 0 && hfdl_ring_size(ring->buf) < (leased + 1) * need && !block_connection_is_shutdown_signaled(block->consumer.in))
						rc = pthread_cond_timedwait(ring->cond, ring->mutex, &until);
					s_grace += now_s() - tg;
					waited_grace = true;
					if (hfdl_ring_size(ring->buf) >= (leased + 1) * need || block_connection_is_shutdown_signaled(block->consumer.in)) continue;
					pthread_mutex_unlock(ring->mutex);
					int32_t n = 0;
					do {                     /* draining collection: the half being filled is closed as it is */
						if (hfdl_gpu_frontend_poll_pdus(fe, pdus, max_pdus, &n) != 0) break;
						for (int32_t i = 0; i < n; i++) push_pdu(&pdus[i], &t0);
						npdus += (uint64_t)n;
					} while (n == max_pdus);
					drains++;
					undelivered = 0;
					pthread_mutex_lock(ring->mutex);
					continue;
				}
				pthread_cond_wait(ring->cond, ring->mutex);
			}
			blk = ok ? hfdl_ring_peek(ring->buf, leased * need, need) : NULL;
			if (ok && blk == NULL) {
				/* a block that wraps around the end of a ring this library did not size: one copy.  The blocks still leased
				 * to the DMA engine sit in front of it; they go back to the producer first. */
				pthread_mutex_unlock(ring->mutex);
				while (ok && leased > 0)
					if (release_copied(fe, ring, need, uploads, &leased, true) != 0) {
						fprintf(stderr, "GPU front end: %s\n", hfdl_gpu_last_error());
						do_exit = 1;
						ok = 0;
					}
				pthread_mutex_lock(ring->mutex);
				if (ok && bounce == NULL) bounce = hfdl_xcalloc(need, sizeof(float complex));
				if (ok && hfdl_ring_read(ring->buf, bounce, need) != need) {      /* only a cf32 ring can be copied out of */
					fprintf(stderr, "GPU front end: input ring holds raw samples in blocks that are not contiguous\n");
					do_exit = 1;
					ok = 0;
				}
				blk = bounce;
				from_bounce = true;
			}
			if (!ok) hfdl_ring_drop(ring->buf, hfdl_ring_size(ring->buf));
			pthread_mutex_unlock(ring->mutex);
			if (!ok) { pthread_cond_signal(ring->cond); continue; }
		}
		const double tw1 = now_s();
		clock_gettime(CLOCK_REALTIME, &last_push);
		if (k == 0) t_first = tw1; else s_wait += tw1 - tw0;
		if (hfdl_gpu_frontend_push_block_raw(fe, blk, need, from_bounce ? HFDL_GPU_SFMT_CF32 : gfmt, 0) != 0) {
			fprintf(stderr, "GPU front end: %s\n", hfdl_gpu_last_error());
			do_exit = 1;
			ok = 0;
			/* nothing of ours may stay with the DMA engine: let the queued copies finish, then give every slot back */
			(void)hfdl_gpu_frontend_prefetch_cancel(fe);
			(void)hfdl_gpu_frontend_input_done(fe);
			leased = 0;
			q_len = 0;
			continue;
		}
		k++;
		undelivered++;
		if (q_len > 0) { q_head = (q_head + 1) % (HFDL_GPU_PREFETCH_MAX + 1); q_len--; }       /* its slot was leased when the copy was queued */
		else { uploads++; if (!from_bounce) leased++; }
		/* queue the uploads of every further whole block the ring holds already: they run beside the blocks still computing */
		while (!from_bounce && q_len < depth) {
			pthread_mutex_lock(ring->mutex);
			const void *next = hfdl_ring_size(ring->buf) >= (leased + 1) * need ? hfdl_ring_peek(ring->buf, leased * need, need) : NULL;
			pthread_mutex_unlock(ring->mutex);
			if (next == NULL || hfdl_gpu_frontend_prefetch_block_raw(fe, next, need, gfmt) != 0) break;
			queued[(q_head + q_len) % (HFDL_GPU_PREFETCH_MAX + 1)] = next;
			q_len++;
			uploads++;
			leased++;
		}
		const double tw2 = now_s();
		s_push += tw2 - tw1;
		/* collect what is known to be complete without draining anything and WITHOUT waiting: what this thread may queue ahead is
		 * bounded by the ring slots it leases to the uploads (it sleeps in release_copied() when the ring has nothing new and the
		 * oldest upload is still running), and slots can only go back to the producer while this thread is not blocked elsewhere */
		int32_t n = 0;
		do {
			if (hfdl_gpu_frontend_poll_pdus_ready(fe, pdus, max_pdus, &n, 2) != 0) break;
			for (int32_t i = 0; i < n; i++) push_pdu(&pdus[i], &t0);
			npdus += (uint64_t)n;
		} while (n == max_pdus);
		const double tw3 = now_s();
		s_poll += tw3 - tw2;
		/* ring slots whose upload has finished go back to the producer; nothing is waited for here */
		if (release_copied(fe, ring, need, uploads, &leased, false) != 0) {
			fprintf(stderr, "GPU front end: %s\n", hfdl_gpu_last_error());
			do_exit = 1;
			ok = 0;
		}
		s_release += now_s() - tw3;
		/* the StatsD counters / gauges are read from the device every 50 ms of wall time at most: one strided device read per
		 * block would cost more than a block of a small geometry takes (a block is decoded in ~0.3 ms) */
		const double now = now_s();
		if (now - t_published >= 0.05) { publish_counters(fe, stats, (int32_t)nch); t_published = now; }
	}
shutdown:
	if (fe) {
		int32_t n = 0;
		do {                                                             /* drain what the lagging collection left behind */
			if (hfdl_gpu_frontend_poll_pdus(fe, pdus, max_pdus, &n) != 0) break;
			for (int32_t i = 0; i < n; i++) push_pdu(&pdus[i], &t0);
			npdus += (uint64_t)n;
		} while (n == max_pdus);
		t_last = now_s();
		publish_counters(fe, stats, (int32_t)nch);
		pthread_mutex_lock(&g_run_lock);
		g_run.blocks = k; g_run.samples = k * (uint64_t)need; g_run.pdus = npdus;
		g_run.seconds = k ? t_last - t_first : 0.0;
		g_run.bytes_per_sample = (int32_t)elem; g_run.channels = (int32_t)nch; g_run.block_samples = (int32_t)need;
		g_run.zero_copy = hfdl_ring_is_pinned(ring->buf);
		g_run.wait_input_s = s_wait; g_run.push_s = s_push; g_run.collect_s = s_poll; g_run.release_s = s_release;
		g_run.drains = drains; g_run.grace_s = s_grace;
		g_run.mpdus_walked = g_lpdu_tally[0]; g_run.lpdus_processed = g_lpdu_tally[1]; g_run.lpdus_good = g_lpdu_tally[2]; g_run.lpdus_bad_fcs = g_lpdu_tally[3];
		pthread_mutex_unlock(&g_run_lock);
	}
	block_connection_one2many_shutdown(down);
	if (fe) hfdl_gpu_frontend_destroy(fe);
	free(bounce);
	free(pdus);
	free(stats);
	free(freqs);
	block->running = false;
	return NULL;
}

size_t hfdl_frontend_block_samples(const struct block *sink)
{
	if (sink == NULL || sink->thread_routine != frontend_thread) return 0;
	return (size_t)container_of(sink, struct gpu_fft_block, block)->geo.input_size;
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
