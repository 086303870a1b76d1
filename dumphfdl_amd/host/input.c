/* input.c -- input registry, sample converters, ring producer and the raw I/Q file input.
 * Interface and behaviour follow src/input-common.c, src/input-helpers.c:10-156 and src/input-file.c:15-119. */
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <stdbool.h>
#include <sys/stat.h>
#include <time.h>
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

struct file_input { struct input input; int fd; bool seekable; };

static int g_file_loops = 1;
void hfdl_file_input_set_loops(int loops) { g_file_loops = loops > 0 ? loops : 1; }

static struct input *file_create(struct input_cfg *cfg)
{
	(void)cfg;
	struct file_input *fi = hfdl_xcalloc(1, sizeof(*fi));
	fi->fd = -1;
	return &fi->input;
}

static void file_destroy(struct input *in)
{
	if (in == NULL) return;
	struct file_input *fi = container_of(in, struct file_input, input);
	if (fi->fd >= 0 && fi->fd != STDIN_FILENO) close(fi->fd);      /* opened by init, thread never ran */
	free(fi);
}

static int32_t file_init(struct input *in)
{
	struct file_input *fi = container_of(in, struct file_input, input);
	struct input_cfg *cfg = in->config;
	if (cfg->sfmt == SFMT_UNDEF) { fprintf(stderr, "Sample format must be specified for file inputs\n"); return -1; }
	if (cfg->read_buffer_size <= 0) cfg->read_buffer_size = FILE_BUFSIZE_DEFAULT;
	fi->fd = strcmp(cfg->source, "-") == 0 ? STDIN_FILENO : open(cfg->source, O_RDONLY);
	if (fi->fd < 0) { fprintf(stderr, "Failed to open input file %s: %s\n", cfg->source, strerror(errno)); return -1; }
	struct stat sb;
	fi->seekable = fstat(fi->fd, &sb) == 0 && S_ISREG(sb.st_mode);
	in->full_scale = get_sample_full_scale_value(cfg->sfmt);
	in->bytes_per_sample = (int32_t)get_sample_size(cfg->sfmt);
	if (cfg->read_buffer_size % in->bytes_per_sample != 0) {
		fprintf(stderr, "Invalid --read-buffer-size value (must be a multiple of sample size, which is %d bytes)\n", in->bytes_per_sample);
		return -1;
	}
	in->block.producer.max_tu = (size_t)(cfg->read_buffer_size / in->bytes_per_sample);
	return 0;
}

static size_t read_fully(int fd, void *buf, size_t want)        /* what fread() does: short only at end of input */
{
	size_t got = 0;
	while (got < want) {
		ssize_t n = read(fd, (char *)buf + got, want - got);
		if (n < 0 && errno == EINTR) continue;
		if (n <= 0) break;
		got += (size_t)n;
	}
	return got;
}

/* ---- parallel positional reads: a regular file is copied out of the page cache by several threads at once ----
 * One thread moves ~3-8 GB/s from the page cache; the front end takes cf32 at up to 32 GB/s (4 Gsamples/s on the small geometries). */
#define FILE_READERS_MAX 16
#define FILE_PIECE (1u << 20)
#define FILE_JOB_MAX ((size_t)64 << 20)
struct read_pool {
	pthread_t th[FILE_READERS_MAX];
	int nthreads;
	pthread_mutex_t lock;
	pthread_cond_t go, done;
	int fd;
	char *dst;
	off_t off;
	size_t len, next;            /* one job at a time: bytes [next, len) are not yet claimed by a worker */
	size_t piece_got[FILE_JOB_MAX / FILE_PIECE];        /* bytes read into piece i ... */
	unsigned char piece_done[FILE_JOB_MAX / FILE_PIECE]; /* ... once its worker is back */
	int busy;                    /* workers inside pread() */
	bool quit;
};

static void *read_worker(void *ctx)
{
	struct read_pool *p = ctx;
	pthread_mutex_lock(&p->lock);
	for (;;) {
		while (!p->quit && p->next >= p->len) pthread_cond_wait(&p->go, &p->lock);
		if (p->quit) break;
		size_t at = p->next, n = p->len - at < FILE_PIECE ? p->len - at : FILE_PIECE;
		p->next += n;
		p->busy++;
		pthread_mutex_unlock(&p->lock);
		size_t got = 0;
		while (got < n) {
			ssize_t k = pread(p->fd, p->dst + at + got, n - got, p->off + (off_t)(at + got));
			if (k < 0 && errno == EINTR) continue;
			if (k <= 0) break;
			got += (size_t)k;
		}
		pthread_mutex_lock(&p->lock);
		p->piece_got[at / FILE_PIECE] = got;
		p->piece_done[at / FILE_PIECE] = 1;
		p->busy--;
		pthread_cond_signal(&p->done);       /* every piece: the caller hands finished pieces on in order, as they come */
	}
	pthread_mutex_unlock(&p->lock);
	return NULL;
}

static void pool_start(struct read_pool *p, int fd)
{
	memset(p, 0, sizeof(*p));
	pthread_mutex_init(&p->lock, NULL);
	pthread_cond_init(&p->go, NULL);
	pthread_cond_init(&p->done, NULL);
	p->fd = fd;
	/* the front end takes cf32 at up to 4 Gsamples/s = 32 GB/s out of the page cache; a thread moves 3-8 GB/s depending on the host */
	long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
	int want = ncpu >= 64 ? FILE_READERS_MAX : ncpu >= 16 ? 8 : ncpu >= 4 ? (int)ncpu / 2 : 1;
	const char *e = getenv("HFDL_FILE_READERS");                  /* tuning / A-B measurements */
	if (e != NULL && atoi(e) >= 1 && atoi(e) <= FILE_READERS_MAX) want = atoi(e);
	for (int i = 0; i < want; i++) if (pthread_create(&p->th[p->nthreads], NULL, read_worker, p) == 0) p->nthreads++;
}

/* Read [off, off+len) of the file (len <= FILE_JOB_MAX) into dst with all workers; returns the bytes read (short at end of file).
 * progress(ctx, n) is called for every n further bytes that are in place IN ORDER from dst, as the pieces come in: the consumer
 * sees the first megabytes of a 60 MB job while the rest is still being read, instead of nothing and then all of it. */
static size_t pool_read(struct read_pool *p, void *dst, off_t off, size_t len, void (*progress)(void *ctx, size_t n), void *ctx)
{
	if (p->nthreads == 0 || len < 2 * FILE_PIECE) {
		size_t got = 0;
		while (got < len) {
			ssize_t k = pread(p->fd, (char *)dst + got, len - got, off + (off_t)got);
			if (k < 0 && errno == EINTR) continue;
			if (k <= 0) break;
			got += (size_t)k;
		}
		if (got) progress(ctx, got);
		return got;
	}
	const size_t npieces = (len + FILE_PIECE - 1) / FILE_PIECE;
	pthread_mutex_lock(&p->lock);
	memset(p->piece_done, 0, npieces);
	p->dst = dst; p->off = off; p->next = 0; p->len = len;
	pthread_cond_broadcast(&p->go);
	size_t got = 0;
	bool short_piece = false;               /* end of file met: what lies behind it in this job is not data */
	for (size_t i = 0; i < npieces; i++) {
		while (!p->piece_done[i]) pthread_cond_wait(&p->done, &p->lock);
		const size_t want = len - i * FILE_PIECE < FILE_PIECE ? len - i * FILE_PIECE : FILE_PIECE, have = short_piece ? 0 : p->piece_got[i];
		if (have) {
			pthread_mutex_unlock(&p->lock);
			progress(ctx, have);
			pthread_mutex_lock(&p->lock);
			got += have;
		}
		if (have < want) short_piece = true;
	}
	while (p->busy > 0) pthread_cond_wait(&p->done, &p->lock);
	p->len = p->next = 0;
	pthread_mutex_unlock(&p->lock);
	return got;
}

static void pool_stop(struct read_pool *p)
{
	pthread_mutex_lock(&p->lock);
	p->quit = true;
	pthread_cond_broadcast(&p->go);
	pthread_mutex_unlock(&p->lock);
	for (int i = 0; i < p->nthreads; i++) pthread_join(p->th[i], NULL);
	pthread_mutex_destroy(&p->lock);
	pthread_cond_destroy(&p->go);
	pthread_cond_destroy(&p->done);
}

/* `n` more bytes are in place behind the pointer hfdl_ring_write_acquire() gave: make the whole samples readable, wake the consumer */
static void commit_bytes(void *ctx, size_t n)
{
	struct circ_buffer *cb = ctx;
	pthread_mutex_lock(cb->mutex);
	hfdl_ring_write_commit(cb->buf, n);
	pthread_mutex_unlock(cb->mutex);
	pthread_cond_signal(cb->cond);
}

/* The ring's element is this file's sample (raw ring in front of the GPU front end, or a cf32 file into any cf32 ring,
 * where convert_cf32 is the identity: x / 1.0f): the file is read STRAIGHT into the ring's free space -- no conversion pass,
 * no bounce buffers.  Regular files are read a front-end block at a time by the reader pool; a pipe delivers what it has. */
static void file_loop_direct(struct input *in, struct file_input *fi, struct circ_buffer *cb)
{
	struct read_pool pool;
	if (fi->seekable) pool_start(&pool, fi->fd);
	off_t off = 0;
	int loops_left = g_file_loops;
	const size_t chunk_max = fi->seekable ? FILE_JOB_MAX : (size_t)in->config->read_buffer_size;
	for (;;) {
		void *dst = NULL;
		size_t room;
		pthread_mutex_lock(cb->mutex);
		/* back-pressure: wait for free space; the front end signals the condition when it releases a block, the timeout
		 * covers consumers that do not (the reference polls with 100 ms naps, src/input-file.c:53-61) */
		/* room == 0 only: with a fragment of a sample carried over (pipe input) the run up to the end of the storage can be
		 * SHORTER than a sample -- exactly the bytes that complete it, after which the tail wraps */
		while ((room = hfdl_ring_write_acquire(cb->buf, &dst)) == 0 && do_exit == 0) {
			struct timespec ts;
			clock_gettime(CLOCK_REALTIME, &ts);
			ts.tv_nsec += 2000000;
			if (ts.tv_nsec >= 1000000000) { ts.tv_sec++; ts.tv_nsec -= 1000000000; }
			pthread_cond_timedwait(cb->cond, cb->mutex, &ts);
		}
		pthread_mutex_unlock(cb->mutex);
		if (do_exit) break;
		size_t want = room < chunk_max ? room : chunk_max;
		size_t got;
		if (fi->seekable) {
			got = pool_read(&pool, dst, off, want, commit_bytes, cb);       /* commits piece by piece */
			off += (off_t)got;
			if (got < want && --loops_left > 0) off = 0;            /* --loop: replay the file from the start */
			else if (got < want) loops_left = 0;
			if (got < want) {
				pthread_mutex_lock(cb->mutex);
				hfdl_ring_discard_partial(cb->buf);                   /* end of file: an incomplete last sample is dropped (whole_samples()) */
				pthread_mutex_unlock(cb->mutex);
			}
		} else {
			ssize_t n;
			do n = read(fi->fd, dst, want); while (n < 0 && errno == EINTR);
			got = n > 0 ? (size_t)n : 0;
			if (got == 0) loops_left = 0;
			commit_bytes(cb, got);
		}
		if (loops_left <= 0) break;
	}
	if (fi->seekable) pool_stop(&pool);
}

/* the reference's loop (src/input-file.c:35-74): read a buffer, wait for room, convert, produce */
static void file_loop_converting(struct input *in, struct file_input *fi, struct circ_buffer *cb)
{
	size_t bufsize = (size_t)in->config->read_buffer_size;
	void *raw = hfdl_xcalloc(bufsize, 1);
	float complex *conv = hfdl_xcalloc(bufsize / (size_t)in->bytes_per_sample, sizeof(float complex));
	size_t len;
	int loops_left = g_file_loops;
	for (;;) {
		len = read_fully(fi->fd, raw, bufsize);
		bool at_end = len < bufsize;                 /* fread() is short only at end of input */
		if (len > 0) {
			for (;;) {          /* back-pressure: poll for ring space, 100 ms naps (src/input-file.c:53-61) */
				pthread_mutex_lock(cb->mutex);
				size_t room = hfdl_ring_space_available(cb->buf);
				pthread_mutex_unlock(cb->mutex);
				if (room * (size_t)in->bytes_per_sample >= len || do_exit) break;
				usleep(100000);
			}
			in->convert_sample_buffer(in, raw, len, conv);
			complex_samples_produce(cb, conv, len / (size_t)in->bytes_per_sample);
		}
		if (do_exit) break;
		if (at_end) {
			if (!fi->seekable || --loops_left <= 0) break;
			if (lseek(fi->fd, 0, SEEK_SET) != 0) break;     /* --loop: replay the file from the start */
		}
	}
	free(raw);
	free(conv);
}

static void *file_thread(void *ctx)
{
	struct block *block = ctx;
	struct input *in = container_of(block, struct input, block);
	struct file_input *fi = container_of(in, struct file_input, input);
	struct circ_buffer *cb = &block->producer.out->circ_buffer;
	if (hfdl_ring_format(cb->buf) == (int)in->config->sfmt) file_loop_direct(in, fi, cb);
	else file_loop_converting(in, fi, cb);
	if (fi->fd != STDIN_FILENO) close(fi->fd);
	fi->fd = -1;
	block_connection_one2one_shutdown(block->producer.out);
	do_exit = 1;
	block->running = false;
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

/* The inputs this library knows, whatever the caller's enum calls them.  The reference's input_type numbering depends on the host
 * program's WITH_SOAPYSDR (src/input-common.h:8-15): INPUT_TYPE_FILE is 1 without it, 2 with it.  This file is compiled without;
 * include/hfdl_host.h binds a WITH_SOAPYSDR caller to the *_with_soapysdr entry points below. */
enum input_kind { KIND_NONE = 0, KIND_FILE, KIND_SOAPYSDR, KIND_MAX };
static struct input_vtable const *g_vtables[KIND_MAX];

static enum input_kind kind_of(int type, bool caller_has_soapysdr)
{
	if (caller_has_soapysdr) return type == 1 ? KIND_SOAPYSDR : type == 2 ? KIND_FILE : KIND_NONE;
	return type == INPUT_TYPE_FILE ? KIND_FILE : KIND_NONE;
}

static int32_t vtable_register(enum input_kind kind, struct input_vtable const *vtable)
{
	if (kind == KIND_NONE || vtable == NULL) return -1;
	if (vtable->create == NULL || vtable->init == NULL || vtable->destroy == NULL || vtable->rx_thread_routine == NULL) return -1;
	g_vtables[kind] = vtable;
	return 0;
}

int32_t input_vtable_register(input_type type, struct input_vtable const *vtable) { return vtable_register(kind_of((int)type, false), vtable); }
int32_t input_vtable_register_with_soapysdr(int type, struct input_vtable const *vtable) { return vtable_register(kind_of(type, true), vtable); }

static struct input_vtable const *input_vtable_get(enum input_kind kind)
{
	if (kind == KIND_NONE) return NULL;
	if (g_vtables[kind] != NULL) return g_vtables[kind];
	return kind == KIND_FILE ? &file_vtable : NULL;                /* SoapySDR: only if the host program registered it */
}

int hfdl_file_input_raw_format(const struct block *source)
{
	if (source == NULL || source->thread_routine != file_thread) return SFMT_CF32;
	const struct input *in = container_of(source, struct input, block);
	return in->config ? (int)in->config->sfmt : SFMT_CF32;
}

static struct block *create_of_kind(struct input_cfg *cfg, enum input_kind kind)
{
	struct input_vtable const *vt = input_vtable_get(kind);
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

struct block *input_create(struct input_cfg *cfg) { return cfg ? create_of_kind(cfg, kind_of((int)cfg->type, false)) : NULL; }
struct block *input_create_with_soapysdr(struct input_cfg *cfg) { return cfg ? create_of_kind(cfg, kind_of((int)cfg->type, true)) : NULL; }

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
