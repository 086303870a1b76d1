/* input.c -- input registry, sample converters, ring producer and the raw I/Q file input.
 * Interface and behaviour follow src/input-common.c, src/input-helpers.c:10-156 and src/input-file.c:15-119. */
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include "hfdl_host.h"
#include "host_internal.h"

#define FILE_BUFSIZE_DEFAULT 320000U     /* src/input-file.c:15 */

/* ---- converters: raw octets -> float complex scaled by 1 / full_scale ---- */

static size_t whole_samples(struct input *in, size_t len) { return len - (len % (size_t)in->bytes_per_sample); }

static void from_cf32(struct input *in, void *buf, size_t len, float complex *out)
{
	const float *f = buf;
	size_t n = whole_samples(in, len) / (2 * sizeof(float));
	for (size_t i = 0; i < n; i++) out[i] = CMPLXF(f[2 * i] / in->full_scale, f[2 * i + 1] / in->full_scale);
}

static void from_cs16(struct input *in, void *buf, size_t len, float complex *out)
{
	const int16_t *s = buf;
	size_t n = whole_samples(in, len) / (2 * sizeof(int16_t));
	for (size_t i = 0; i < n; i++) out[i] = CMPLXF((float)s[2 * i] / in->full_scale, (float)s[2 * i + 1] / in->full_scale);
}

static void from_cu8(struct input *in, void *buf, size_t len, float complex *out)
{
	const uint8_t *b = buf;
	size_t n = whole_samples(in, len) / 2;
	const float shift = in->full_scale / 2.0f;       /* as written in the reference, src/input-helpers.c:71 */
	for (size_t i = 0; i < n; i++) out[i] = CMPLXF((b[2 * i] - shift) / in->full_scale, (b[2 * i + 1] - shift) / in->full_scale);
}

static const struct { const char *name; size_t size; float full_scale; convert_sample_buffer_fun fn; } formats[SFMT_MAX] = {
	[SFMT_UNDEF] = { "", 0, 0.f, NULL },
	[SFMT_CU8] = { "CU8", 2, (float)SCHAR_MAX, from_cu8 },
	[SFMT_CS16] = { "CS16", 4, (float)SHRT_MAX + 0.5f, from_cs16 },
	[SFMT_CF32] = { "CF32", 8, 1.0f, from_cf32 },
};

size_t get_sample_size(sample_format f) { return f < SFMT_MAX ? formats[f].size : 0; }
float get_sample_full_scale_value(sample_format f) { return f < SFMT_MAX ? formats[f].full_scale : 0.f; }
convert_sample_buffer_fun get_sample_converter(sample_format f) { return f < SFMT_MAX ? formats[f].fn : NULL; }

sample_format sample_format_from_string(char const *str)
{
	if (str == NULL) return SFMT_UNDEF;
	for (int f = SFMT_UNDEF + 1; f < SFMT_MAX; f++) if (strcasecmp(str, formats[f].name) == 0) return (sample_format)f;
	return SFMT_UNDEF;
}

void complex_samples_produce(struct circ_buffer *cb, float complex *samples, size_t num_samples)
{
	pthread_mutex_lock(cb->mutex);
	size_t room = hfdl_ring_space_available(cb->buf);
	if (room < num_samples) {
		fprintf(stderr, "Sample buffer overrun (%zu/%zu samples lost)\n", num_samples - room, num_samples);
		num_samples = room;
	}
	hfdl_ring_write(cb->buf, samples, num_samples);
	pthread_mutex_unlock(cb->mutex);
	pthread_cond_signal(cb->cond);
}

/* ---- file input ---- */

struct file_input { struct input input; FILE *fh; };

static struct input *file_create(struct input_cfg *cfg)
{
	(void)cfg;
	struct file_input *fi = hfdl_xcalloc(1, sizeof(*fi));
	return &fi->input;
}

static void file_destroy(struct input *in)
{
	if (in) free(container_of(in, struct file_input, input));
}

static int32_t file_init(struct input *in)
{
	struct file_input *fi = container_of(in, struct file_input, input);
	struct input_cfg *cfg = in->config;
	if (cfg->sfmt == SFMT_UNDEF) { fprintf(stderr, "Sample format must be specified for file inputs\n"); return -1; }
	if (cfg->read_buffer_size <= 0) cfg->read_buffer_size = FILE_BUFSIZE_DEFAULT;
	fi->fh = strcmp(cfg->source, "-") == 0 ? stdin : fopen(cfg->source, "rb");
	if (fi->fh == NULL) { fprintf(stderr, "Failed to open input file %s: %s\n", cfg->source, strerror(errno)); return -1; }
	in->full_scale = get_sample_full_scale_value(cfg->sfmt);
	in->bytes_per_sample = (int32_t)get_sample_size(cfg->sfmt);
	if (cfg->read_buffer_size % in->bytes_per_sample != 0) {
		fprintf(stderr, "Invalid --read-buffer-size value (must be a multiple of sample size, which is %d bytes)\n", in->bytes_per_sample);
		return -1;
	}
	in->block.producer.max_tu = (size_t)(cfg->read_buffer_size / in->bytes_per_sample);
	return 0;
}

static void *file_thread(void *ctx)
{
	struct block *block = ctx;
	struct input *in = container_of(block, struct input, block);
	struct file_input *fi = container_of(in, struct file_input, input);
	struct circ_buffer *cb = &block->producer.out->circ_buffer;
	size_t bufsize = (size_t)in->config->read_buffer_size;
	void *raw = hfdl_xcalloc(bufsize, 1);
	float complex *conv = hfdl_xcalloc(bufsize / (size_t)in->bytes_per_sample, sizeof(float complex));
	size_t len;
	do {
		len = fread(raw, 1, bufsize, fi->fh);
		for (;;) {              /* back-pressure: poll for ring space, 100 ms naps (src/input-file.c:53-61) */
			pthread_mutex_lock(cb->mutex);
			size_t room = hfdl_ring_space_available(cb->buf);
			pthread_mutex_unlock(cb->mutex);
			if (room * (size_t)in->bytes_per_sample >= len) break;
			usleep(100000);
		}
		in->convert_sample_buffer(in, raw, len, conv);
		complex_samples_produce(cb, conv, len / (size_t)in->bytes_per_sample);
	} while (len > 0 && do_exit == 0);
	if (fi->fh != stdin) fclose(fi->fh);
	fi->fh = NULL;
	block_connection_one2one_shutdown(block->producer.out);
	do_exit = 1;
	block->running = false;
	free(raw);
	free(conv);
	return NULL;
}

static struct input_vtable file_vtable = { file_create, file_init, file_destroy, file_thread };

/* ---- registry (src/input-common.c) ---- */

struct input_cfg *input_cfg_create(void)
{
	struct input_cfg *cfg = hfdl_xcalloc(1, sizeof(*cfg));
	cfg->sfmt = SFMT_UNDEF;
	cfg->type = INPUT_TYPE_UNDEF;
	return cfg;
}

void input_cfg_destroy(struct input_cfg *cfg) { free(cfg); }

static struct input_vtable const *g_vtables[INPUT_TYPE_MAX];

int32_t input_vtable_register(input_type type, struct input_vtable const *vtable)
{
	if (type <= INPUT_TYPE_UNDEF || type >= INPUT_TYPE_MAX || vtable == NULL) return -1;
	if (vtable->create == NULL || vtable->init == NULL || vtable->destroy == NULL || vtable->rx_thread_routine == NULL) return -1;
	g_vtables[type] = vtable;
	return 0;
}

static struct input_vtable const *input_vtable_get(input_type type)
{
	if (type <= INPUT_TYPE_UNDEF || type >= INPUT_TYPE_MAX) return NULL;
	if (g_vtables[type] != NULL) return g_vtables[type];
	return type == INPUT_TYPE_FILE ? &file_vtable : NULL;          /* SoapySDR: only if the host program registered it */
}

struct block *input_create(struct input_cfg *cfg)
{
	if (cfg == NULL) return NULL;
	struct input_vtable const *vt = input_vtable_get(cfg->type);
	if (vt == NULL) return NULL;
	struct input *in = vt->create(cfg);
	if (in == NULL) return NULL;
	in->vtable = (struct input_vtable *)vt;
	in->config = cfg;
	in->block.producer.type = PRODUCER_SINGLE;
	in->block.producer.max_tu = 0;
	in->block.consumer.type = CONSUMER_NONE;
	in->block.thread_routine = vt->rx_thread_routine;
	return &in->block;
}

int32_t input_init(struct block *block)
{
	if (block == NULL) return -1;
	struct input *in = container_of(block, struct input, block);
	int32_t rc = in->vtable->init(in);
	if (rc < 0) return rc;
	in->convert_sample_buffer = get_sample_converter(in->config->sfmt);
	if (in->convert_sample_buffer == NULL || in->bytes_per_sample <= 0 || in->full_scale <= 0.f || block->producer.max_tu == 0) return -1;
	return 0;
}

void input_destroy(struct block *block)
{
	if (block == NULL) return;
	struct input *in = container_of(block, struct input, block);
	in->vtable->destroy(in);
}
