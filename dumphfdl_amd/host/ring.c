/* ring.c -- sample FIFO behind struct circ_buffer (the reference uses liquid-dsp's cbuffercf: src/block.c:20, src/fft.c:41-54,
 * src/input-helpers.c:83-89).  Single producer / single consumer; callers hold the connection mutex as in the reference.
 *
 * Two things the reference's ring does not have, both to keep host passes over the samples off the path to the GPU:
 *   - the storage can be PAGE-LOCKED and a whole number of front-end blocks long, so the GPU front end DMAs a block straight
 *     out of the ring (hfdl_ring_peek / hfdl_ring_drop) instead of copying it into a staging buffer first;
 *   - the element can be a RAW sample (cs16 = 4 bytes, cu8 = 2 bytes) instead of float complex: the library's own file
 *     input then reads the file straight into the ring (hfdl_ring_write_acquire / _commit) and the converter of
 *     src/input-helpers.c:33-78 runs on the device inside the forward FFT's first load.
 * A ring made by hfdl_ring_create() is the plain cf32 FIFO the reference's API implies. */
#include <stdlib.h>
#include <string.h>
#include "hfdl_host.h"
#include "hfdl_gpu.h"
#include "host_internal.h"

struct hfdl_ring {
	unsigned char *data;
	size_t cap, head, count;       /* in samples; head = index of the oldest sample */
	size_t elem;                   /* bytes per sample: 8 (float complex), 4 (cs16) or 2 (cu8) */
	size_t carry;                  /* bytes of an incomplete sample already written after the tail (pipe input) */
	int fmt;                       /* sample_format of the elements (SFMT_CF32 for the classic ring) */
	int pinned;                    /* storage came from hfdl_gpu_host_alloc() */
};

struct hfdl_ring *hfdl_ring_create_ex(size_t capacity, int fmt, int want_pinned)
{
	struct hfdl_ring *r = hfdl_xcalloc(1, sizeof(*r));
	r->elem = fmt == SFMT_CS16 ? 4 : fmt == SFMT_CU8 ? 2 : sizeof(float complex);
	r->fmt = fmt == SFMT_CS16 || fmt == SFMT_CU8 ? fmt : SFMT_CF32;
	r->cap = capacity;
	size_t bytes = (capacity ? capacity : 1) * r->elem;
	void *p = NULL;
	if (want_pinned && hfdl_gpu_host_alloc(&p, bytes) == 0 && p != NULL) {
		memset(p, 0, bytes);
		r->pinned = 1;
	} else {
		p = hfdl_xcalloc(bytes, 1);
	}
	r->data = p;
	return r;
}

struct hfdl_ring *hfdl_ring_create(size_t capacity) { return hfdl_ring_create_ex(capacity, SFMT_CF32, 0); }

void hfdl_ring_destroy(struct hfdl_ring *r)
{
	if (r == NULL) return;
	if (r->pinned) hfdl_gpu_host_free(r->data); else free(r->data);
	free(r);
}

size_t hfdl_ring_size(const struct hfdl_ring *r) { return r->count; }
size_t hfdl_ring_space_available(const struct hfdl_ring *r) { return r->cap - r->count; }
size_t hfdl_ring_capacity(const struct hfdl_ring *r) { return r->cap; }
size_t hfdl_ring_elem_size(const struct hfdl_ring *r) { return r->elem; }
int    hfdl_ring_format(const struct hfdl_ring *r) { return r->fmt; }
int    hfdl_ring_is_pinned(const struct hfdl_ring *r) { return r->pinned; }

static size_t ring_write_bytes(struct hfdl_ring *r, const void *src, size_t n)
{
	if (n > r->cap - r->count) n = r->cap - r->count;
	if (n == 0) return 0;                          /* full, or a ring of capacity 0 */
	size_t tail = (r->head + r->count) % r->cap;
	size_t first = n < r->cap - tail ? n : r->cap - tail;
	memcpy(r->data + tail * r->elem, src, first * r->elem);
	memcpy(r->data, (const unsigned char *)src + first * r->elem, (n - first) * r->elem);
	r->count += n;
	return n;
}

size_t hfdl_ring_write(struct hfdl_ring *r, const float complex *src, size_t n)
{
	if (r->elem != sizeof(float complex)) return 0;        /* a raw ring takes raw samples only (hfdl_ring_write_acquire) */
	return ring_write_bytes(r, src, n);
}

size_t hfdl_ring_read(struct hfdl_ring *r, float complex *dst, size_t n)
{
	if (r->elem != sizeof(float complex)) return 0;
	if (n > r->count) n = r->count;
	if (n == 0) return 0;
	size_t first = n < r->cap - r->head ? n : r->cap - r->head;
	memcpy(dst, r->data + r->head * r->elem, first * r->elem);
	memcpy(dst + first, r->data, (n - first) * r->elem);
	r->head = (r->head + n) % r->cap;
	r->count -= n;
	return n;
}

/* ---- zero-copy producer side: fill the free space in place ---- */

/* The longest contiguous free run after the tail, in BYTES, and where it starts (an incomplete sample left by the previous
 * commit sits right before it).  The caller writes up to *bytes there without the lock held (single producer) and then
 * commits under the lock.  Returns 0 when the ring is full. */
size_t hfdl_ring_write_acquire(struct hfdl_ring *r, void **ptr)
{
	size_t free_samples = r->cap - r->count;
	if (free_samples == 0) { *ptr = NULL; return 0; }
	size_t tail = (r->head + r->count) % r->cap;
	size_t run = free_samples < r->cap - tail ? free_samples : r->cap - tail;
	*ptr = r->data + tail * r->elem + r->carry;
	return run * r->elem - r->carry;
}

/* `bytes` more bytes are in place after the pointer hfdl_ring_write_acquire() gave: whole samples become readable, an
 * incomplete trailing sample is carried to the next commit.  Returns the samples made readable. */
size_t hfdl_ring_write_commit(struct hfdl_ring *r, size_t bytes)
{
	size_t total = r->carry + bytes;
	size_t n = total / r->elem;
	r->carry = total % r->elem;
	r->count += n;
	return n;
}

void hfdl_ring_discard_partial(struct hfdl_ring *r) { r->carry = 0; }

/* ---- zero-copy consumer side: look at samples in place, release them later ---- */

/* pointer to `n` contiguous readable samples starting `offset` samples after the oldest one, or NULL if they are not all
 * there or wrap around the end of the storage (never, when capacity and every read are multiples of the same block size) */
const void *hfdl_ring_peek(const struct hfdl_ring *r, size_t offset, size_t n)
{
	if (offset + n > r->count || r->cap == 0) return NULL;
	size_t at = (r->head + offset) % r->cap;
	if (at + n > r->cap) return NULL;
	return r->data + at * r->elem;
}

size_t hfdl_ring_drop(struct hfdl_ring *r, size_t n)
{
	if (n > r->count) n = r->count;
	if (n == 0) return 0;
	r->head = (r->head + n) % r->cap;
	r->count -= n;
	return n;
}
