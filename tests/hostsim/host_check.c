/* host_check.c -- TEST PROGRAM: drives libhfdl_host.so's block / input API the way dumphfdl's main() does, without a GPU.
 *   host_check ring                       ring wrap-around / overrun behaviour
 *   host_check file PATH FMT BUFSIZE OUT  file input -> ring -> this consumer; converted cf32 samples written to OUT
 *   host_check graph                      connect/return-value contract of the block graph and channel slots
 *   host_check plugin                     an input registered with input_vtable_register() (the SoapySDR slot) feeds the ring
 *   host_check direct PATH FMT OUT LOOPS  file input -> ring in front of the GPU front-end block (fft_create(), never started):
 *                                         the ring carries the file's RAW samples, this consumer takes them in place */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hfdl_host.h"
#include "host_internal.h"       /* the in-place ring accessors the front-end block uses (library-private) */

static int check_ring(void)
{
	struct hfdl_ring *r = hfdl_ring_create(8);
	float complex a[16], b[16];
	for (int i = 0; i < 16; i++) a[i] = i + 1 + (float)i * I;
	if (hfdl_ring_write(r, a, 5) != 5 || hfdl_ring_size(r) != 5 || hfdl_ring_space_available(r) != 3) return 1;
	if (hfdl_ring_read(r, b, 3) != 3 || crealf(b[0]) != 1 || crealf(b[2]) != 3) return 2;
	if (hfdl_ring_write(r, a + 5, 6) != 6 || hfdl_ring_size(r) != 8) return 3;          /* wraps */
	if (hfdl_ring_write(r, a, 4) != 0) return 4;                                          /* full */
	if (hfdl_ring_read(r, b, 16) != 8) return 5;
	for (int i = 0; i < 8; i++) if (crealf(b[i]) != i + 4 || cimagf(b[i]) != i + 3) return 6;
	hfdl_ring_destroy(r);
	/* complex_samples_produce drops what does not fit and keeps the rest (src/input-helpers.c:80-92) */
	struct block src = { .producer = { .type = PRODUCER_SINGLE, .max_tu = 1 } }, dst = { .consumer = { .type = CONSUMER_SINGLE, .min_ru = 2 } };
	if (block_connect_one2one(&src, &dst) != 1) return 7;
	struct circ_buffer *cb = &src.producer.out->circ_buffer;           /* capacity max(8*1, 2*2) = 8 */
	complex_samples_produce(cb, a, 6);
	complex_samples_produce(cb, a + 6, 6);
	if (hfdl_ring_size(cb->buf) != 8) return 8;
	hfdl_ring_read(cb->buf, b, 8);
	if (crealf(b[7]) != 8) return 9;
	block_disconnect_one2one(&src, &dst);
	if (src.producer.out != NULL || dst.consumer.in != NULL) return 10;
	printf("ring ok\n");
	return 0;
}

static int check_file(const char *path, const char *fmt, int bufsize, const char *out_path, int loops)
{
	hfdl_file_input_set_loops(loops);
	struct input_cfg *cfg = input_cfg_create();
	cfg->type = INPUT_TYPE_FILE;
	cfg->source = (char *)path;
	cfg->sfmt = sample_format_from_string(fmt);
	cfg->read_buffer_size = bufsize;
	cfg->sample_rate = 250000;
	struct block *in = input_create(cfg);
	if (in == NULL) return 1;
	if (input_init(in) < 0) return 2;
	struct block sink = { .consumer = { .type = CONSUMER_SINGLE, .min_ru = 1000 } };
	if (block_connect_one2one(in, &sink) != 1) return 3;
	struct circ_buffer *cb = &sink.consumer.in->circ_buffer;
	FILE *out = fopen(out_path, "wb");
	if (block_start(in) != 1) return 4;
	float complex tmp[4096];
	size_t total = 0;
	for (;;) {
		pthread_mutex_lock(cb->mutex);
		while (hfdl_ring_size(cb->buf) == 0 && !block_connection_is_shutdown_signaled(sink.consumer.in)) pthread_cond_wait(cb->cond, cb->mutex);
		size_t n = hfdl_ring_read(cb->buf, tmp, 4096);
		int done = n == 0 && block_connection_is_shutdown_signaled(sink.consumer.in);
		pthread_mutex_unlock(cb->mutex);
		if (done) break;
		fwrite(tmp, sizeof(float complex), n, out);
		total += n;
	}
	fclose(out);
	while (block_is_running(in)) usleep(1000);
	struct input *ip = (struct input *)in;
	printf("samples %zu max_tu %zu bytes_per_sample %d full_scale %.3f\n", total, in->producer.max_tu, ip->bytes_per_sample, ip->full_scale);
	block_disconnect_one2one(in, &sink);
	input_destroy(in);
	input_cfg_destroy(cfg);
	return 0;
}

static int check_direct(const char *path, const char *fmt, const char *out_path, int loops)
{
	struct input_cfg *cfg = input_cfg_create();
	cfg->type = INPUT_TYPE_FILE;
	cfg->source = (char *)path;
	cfg->sfmt = sample_format_from_string(fmt);
	cfg->sample_rate = 250000;
	hfdl_file_input_set_loops(loops);
	struct block *in = input_create(cfg);
	if (in == NULL || input_init(in) < 0) return 1;
	struct block *fft = fft_create(compute_fft_decimation_rate(250000, 5400), compute_filter_relative_transition_bw(250000, 250));
	if (fft == NULL) return 2;
	if (block_connect_one2one(in, fft) != 1) return 3;
	struct circ_buffer *cb = &fft->consumer.in->circ_buffer;
	const size_t blk = hfdl_frontend_block_samples(fft), elem = hfdl_ring_elem_size(cb->buf);
	if (blk != 28672 || hfdl_ring_capacity(cb->buf) % blk != 0 || hfdl_ring_capacity(cb->buf) < 6 * blk) return 4;
	if (hfdl_ring_format(cb->buf) != (int)cfg->sfmt || elem != get_sample_size(cfg->sfmt)) return 5;
	float complex one = 1;
	if (elem != 8 && hfdl_ring_write(cb->buf, &one, 1) != 0) return 6;      /* a raw ring refuses cf32 writes */
	FILE *out = fopen(out_path, "wb");
	if (block_start(in) != 1) return 7;
	size_t total = 0, blocks = 0;
	for (;;) {
		pthread_mutex_lock(cb->mutex);
		while (hfdl_ring_size(cb->buf) < blk && !block_connection_is_shutdown_signaled(fft->consumer.in)) pthread_cond_wait(cb->cond, cb->mutex);
		size_t have = hfdl_ring_size(cb->buf);
		size_t take = have >= blk ? blk : have;             /* the tail shorter than a block (the front end leaves it unread) */
		const void *p = take ? hfdl_ring_peek(cb->buf, 0, take) : NULL;
		int done = have < blk && block_connection_is_shutdown_signaled(fft->consumer.in);
		pthread_mutex_unlock(cb->mutex);
		if (take && p == NULL) return 8;                     /* whole blocks never wrap */
		if (take) fwrite(p, elem, take, out);
		pthread_mutex_lock(cb->mutex);
		hfdl_ring_drop(cb->buf, take);
		pthread_mutex_unlock(cb->mutex);
		pthread_cond_signal(cb->cond);
		total += take;
		if (take == blk) blocks++;
		if (done) break;
	}
	fclose(out);
	while (block_is_running(in)) usleep(1000);
	printf("samples %zu blocks %zu elem %zu capacity %zu pinned %d\n", total, blocks, elem, hfdl_ring_capacity(cb->buf), hfdl_ring_is_pinned(cb->buf));
	block_disconnect_one2one(in, fft);
	fft_destroy(fft);
	input_destroy(in);
	input_cfg_destroy(cfg);
	return 0;
}

static int check_graph(void)
{
	int32_t dec = compute_fft_decimation_rate(250000, HFDL_SYMBOL_RATE * SPS);
	float tbw = compute_filter_relative_transition_bw(250000, HFDL_CHANNEL_TRANSITION_BW_HZ);
	if (dec != 32) return 1;
	if (compute_fft_decimation_rate(8000000, 5400) != 1024 || compute_fft_decimation_rate(40000000, 5400) != 4096) return 2;
	struct block *fft = fft_create(dec, tbw);
	if (fft == NULL) return 3;
	if (fft->producer.type != PRODUCER_MULTI || fft->consumer.type != CONSUMER_SINGLE || fft->producer.max_tu != 32768 || fft->consumer.min_ru != 32768) return 4;
	hfdl_init_globals();
	struct block *ch[3];
	for (int i = 0; i < 3; i++) {
		ch[i] = hfdl_channel_create(250000, dec, tbw, 10000000, 10010000 + 20000 * i);
		if (ch[i] == NULL || ch[i]->consumer.type != CONSUMER_MULTI || ch[i]->producer.type != PRODUCER_NONE) return 5;
	}
	if (hfdl_channel_create(0, dec, tbw, 1, 2) != NULL) return 6;
	if (block_connect_one2many(fft, 3, ch) != 3) return 7;
	for (int i = 0; i < 3; i++) if (ch[i]->consumer.in != fft->producer.out) return 8;
	struct block bad = { .producer = { .type = PRODUCER_SINGLE, .max_tu = 4 } };
	if (block_connect_one2many(&bad, 3, ch) != 0) return 9;                 /* wrong producer type */
	if (block_is_running(fft) || block_set_is_any_running(3, ch)) return 10;
	block_disconnect_one2many(fft, 3, ch);
	for (int i = 0; i < 3; i++) { if (ch[i]->consumer.in != NULL) return 11; hfdl_channel_destroy(ch[i]); }
	fft_destroy(fft);
	if (sample_format_from_string("cs16") != SFMT_CS16 || sample_format_from_string("CF32") != SFMT_CF32 ||
			sample_format_from_string("nope") != SFMT_UNDEF) return 12;
	if (get_sample_size(SFMT_CU8) != 2 || get_sample_size(SFMT_CS16) != 4 || get_sample_size(SFMT_CF32) != 8) return 13;
	struct octet_string *o = octet_string_new(malloc(4), 4);
	struct metadata *m = hfdl_pdu_metadata_create();
	if (o->len != 4 || m->vtable == NULL) return 14;
	struct metadata *c = m->vtable->copy(m);
	m->vtable->destroy(m);
	c->vtable->destroy(c);
	octet_string_destroy(o);
	printf("graph ok\n");
	return 0;
}

/* a stand-in for a radio input: produces a ramp in bursts of max_tu samples, like a SoapySDR rx thread would (src/input-soapysdr.c:226-273) */
struct ramp_input { struct input input; int total; };
static struct input *ramp_create(struct input_cfg *cfg) { (void)cfg; struct ramp_input *r = calloc(1, sizeof(*r)); return &r->input; }
static int32_t ramp_init(struct input *in)
{
	in->full_scale = 1.0f; in->bytes_per_sample = 8; in->block.producer.max_tu = 500;
	((struct ramp_input *)in)->total = 5000;
	return 0;
}
static void ramp_destroy(struct input *in) { free(in); }
static void *ramp_thread(void *ctx)
{
	struct block *block = ctx;
	struct ramp_input *r = (struct ramp_input *)block;          /* struct input starts with its block */
	struct circ_buffer *cb = &block->producer.out->circ_buffer;
	float complex buf[500];
	for (int done = 0; done < r->total; done += 500) {
		for (int i = 0; i < 500; i++) buf[i] = (float)(done + i) - (float)(done + i) * I;
		for (;;) {                                               /* wait for room like file_input_thread does */
			pthread_mutex_lock(cb->mutex);
			size_t room = hfdl_ring_space_available(cb->buf);
			pthread_mutex_unlock(cb->mutex);
			if (room >= 500) break;
			usleep(1000);
		}
		complex_samples_produce(cb, buf, 500);
	}
	block_connection_one2one_shutdown(block->producer.out);
	block->running = false;
	return NULL;
}
static struct input_vtable const ramp_vtable = { ramp_create, ramp_init, ramp_destroy, ramp_thread };

static int check_plugin(void)
{
#ifndef WITH_SOAPYSDR
	/* a default build of the host program has no radio slot (src/input-common.h:8-15): nothing beyond INPUT_TYPE_FILE can be registered */
	if (INPUT_TYPE_FILE != 1 || INPUT_TYPE_MAX != 2) return 20;
	if (input_vtable_register(INPUT_TYPE_MAX, &ramp_vtable) == 0 || input_vtable_register((input_type)3, &ramp_vtable) == 0) return 21;
	struct input_cfg *c0 = input_cfg_create();
	c0->type = (input_type)2;                              /* what INPUT_TYPE_FILE would be in the other numbering: not an input here */
	if (input_create(c0) != NULL) return 22;
	input_cfg_destroy(c0);
	printf("plugin ok (no slot in this build)\n");
	return 0;
#else
	if (INPUT_TYPE_SOAPYSDR != 1 || INPUT_TYPE_FILE != 2 || INPUT_TYPE_MAX != 3) return 20;
	struct input_cfg *cfg = input_cfg_create();
	cfg->type = INPUT_TYPE_SOAPYSDR;
	cfg->sfmt = SFMT_CF32;
	cfg->sample_rate = 250000;
	if (input_create(cfg) != NULL) return 1;                                     /* nothing registered: no such input */
	if (input_vtable_register(INPUT_TYPE_MAX, &ramp_vtable) == 0) return 2;
	if (input_vtable_register(INPUT_TYPE_SOAPYSDR, NULL) == 0) return 3;
	if (input_vtable_register(INPUT_TYPE_SOAPYSDR, &ramp_vtable) != 0) return 4;
	struct block *in = input_create(cfg);
	if (in == NULL || input_init(in) < 0) return 5;
	struct block sink = { .consumer = { .type = CONSUMER_SINGLE, .min_ru = 256 } };
	if (block_connect_one2one(in, &sink) != 1) return 6;
	struct circ_buffer *cb = &sink.consumer.in->circ_buffer;
	if (block_start(in) != 1) return 7;
	float complex tmp[1024];
	int seen = 0;
	for (;;) {
		pthread_mutex_lock(cb->mutex);
		while (hfdl_ring_size(cb->buf) == 0 && !block_connection_is_shutdown_signaled(sink.consumer.in)) pthread_cond_wait(cb->cond, cb->mutex);
		size_t n = hfdl_ring_read(cb->buf, tmp, 1024);
		int done = n == 0 && block_connection_is_shutdown_signaled(sink.consumer.in);
		pthread_mutex_unlock(cb->mutex);
		for (size_t i = 0; i < n; i++, seen++) if (crealf(tmp[i]) != (float)seen || cimagf(tmp[i]) != -(float)seen) return 8;
		if (done) break;
	}
	if (seen != 5000) return 9;
	while (block_is_running(in)) usleep(1000);
	block_disconnect_one2one(in, &sink);
	input_destroy(in);
	input_cfg_destroy(cfg);
	printf("plugin ok\n");
	return 0;
#endif
}

int main(int argc, char **argv)
{
	int rc = 99;
	if (argc >= 2 && !strcmp(argv[1], "ring")) rc = check_ring();
	else if (argc >= 6 && !strcmp(argv[1], "file")) rc = check_file(argv[2], argv[3], atoi(argv[4]), argv[5], argc >= 7 ? atoi(argv[6]) : 1);
	else if (argc >= 2 && !strcmp(argv[1], "graph")) rc = check_graph();
	else if (argc >= 2 && !strcmp(argv[1], "plugin")) rc = check_plugin();
	else if (argc >= 6 && !strcmp(argv[1], "direct")) rc = check_direct(argv[2], argv[3], argv[4], atoi(argv[5]));
	if (rc) fprintf(stderr, "host_check %s failed at step %d\n", argc > 1 ? argv[1] : "?", rc);
	return rc;
}
