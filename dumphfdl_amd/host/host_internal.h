/* host_internal.h -- helpers private to libhfdl_host.so */
#pragma once
#include <signal.h>
#include <stddef.h>
#include <stdint.h>
#include "hfdl_host.h"

void *hfdl_xcalloc(size_t nmemb, size_t size);                 /* calloc or _exit(1): src/util.c:25-33 */
int   hfdl_start_detached(pthread_t *th, void *(*fn)(void *), void *ctx);   /* src/util.c:45-63 */
extern volatile sig_atomic_t do_exit;                           /* src/globals.h */

/* channel registry: hfdl_channel_create() -> consumed by the front-end thread */
struct hfdl_channel_slot {
	struct block block;
	int32_t sample_rate, pre_decimation_rate, centerfreq, frequency;
	float transition_bw;
};
size_t hfdl_channels_on_connection(struct block_connection *conn, struct hfdl_channel_slot **out, size_t max);

/* ring extensions (ring.c): raw-sample elements, page-locked storage, in-place producer / consumer access */
struct hfdl_ring *hfdl_ring_create_ex(size_t capacity, int fmt, int want_pinned);
size_t hfdl_ring_capacity(const struct hfdl_ring *r);
size_t hfdl_ring_elem_size(const struct hfdl_ring *r);
int    hfdl_ring_format(const struct hfdl_ring *r);
int    hfdl_ring_is_pinned(const struct hfdl_ring *r);
size_t hfdl_ring_write_acquire(struct hfdl_ring *r, void **ptr);
size_t hfdl_ring_write_commit(struct hfdl_ring *r, size_t bytes);
void   hfdl_ring_discard_partial(struct hfdl_ring *r);
const void *hfdl_ring_peek(const struct hfdl_ring *r, size_t offset, size_t n);
size_t hfdl_ring_drop(struct hfdl_ring *r, size_t n);

/* what block_connect_one2one() asks its two ends (0 / SFMT_CF32 for blocks that are not the library's own) */
size_t hfdl_frontend_block_samples(const struct block *sink);      /* input_size if `sink` is the GPU front-end block, else 0 */
int    hfdl_file_input_raw_format(const struct block *source);     /* the file's sample_format if `source` is the file input, else SFMT_CF32 */

/* the entry points a WITH_SOAPYSDR host program is bound to (include/hfdl_host.h): the caller's numbering has the radio slot at 1 */
#ifndef WITH_SOAPYSDR       /* (a WITH_SOAPYSDR unit already sees them under their public names) */
struct block *input_create_with_soapysdr(struct input_cfg *cfg);
int32_t input_vtable_register_with_soapysdr(int type, struct input_vtable const *vtable);
#endif

#define container_of(ptr, type, member) ((type *)((char *)(ptr) - offsetof(type, member)))
