/* ring.c -- cf32 FIFO behind struct circ_buffer (the reference uses liquid-dsp's cbuffercf: src/block.c:20, src/fft.c:41-54,
 * src/input-helpers.c:83-89).  Single producer / single consumer; callers hold the connection mutex as in the reference. */
#include <stdlib.h>
#include <string.h>
#include "hfdl_host.h"
#include "host_internal.h"

struct hfdl_ring {
	float complex *data;
	size_t cap, head, count;       /* head = index of the oldest sample */
};

struct hfdl_ring *hfdl_ring_create(size_t capacity)
{
	struct hfdl_ring *r = hfdl_xcalloc(1, sizeof(*r));
	r->data = hfdl_xcalloc(capacity ? capacity : 1, sizeof(float complex));
	r->cap = capacity;
	return r;
}

void hfdl_ring_destroy(struct hfdl_ring *r)
{
	if (r == NULL) return;
	free(r->data);
	free(r);
}

size_t hfdl_ring_size(const struct hfdl_ring *r) { return r->count; }
size_t hfdl_ring_space_available(const struct hfdl_ring *r) { return r->cap - r->count; }

size_t hfdl_ring_write(struct hfdl_ring *r, const float complex *src, size_t n)
{
	if (n > r->cap - r->count) n = r->cap - r->count;
	size_t tail = (r->head + r->count) % r->cap;
	size_t first = n < r->cap - tail ? n : r->cap - tail;
	memcpy(r->data + tail, src, first * sizeof(float complex));
	memcpy(r->data, src + first, (n - first) * sizeof(float complex));
	r->count += n;
	return n;
}

size_t hfdl_ring_read(struct hfdl_ring *r, float complex *dst, size_t n)
{
	if (n > r->count) n = r->count;
	size_t first = n < r->cap - r->head ? n : r->cap - r->head;
	memcpy(dst, r->data + r->head, first * sizeof(float complex));
	memcpy(dst + first, r->data, (n - first) * sizeof(float complex));
	r->head = (r->head + n) % r->cap;
	r->count -= n;
	return n;
}
