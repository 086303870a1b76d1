/* blocks.c -- thread-per-block dataflow runtime with the reference's interface (src/block.c:55-193). */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "hfdl_host.h"
#include "hfdl_gpu.h"
#include "host_internal.h"

#define PROD_MTU_MULT 8          /* ring = max(8 x producer MTU, 2 x consumer MRU): src/block.c:15-16,62-64 */
#define CONS_MRU_MULT 2

volatile sig_atomic_t do_exit = 0;

void *hfdl_xcalloc(size_t nmemb, size_t size)
{
	void *p = calloc(nmemb, size);
	if (p == NULL) {
		fprintf(stderr, "calloc(%zu, %zu) failed, aborting\n", nmemb, size);
		_exit(1);
	}
	return p;
}

int hfdl_start_detached(pthread_t *th, void *(*fn)(void *), void *ctx)
{
	pthread_attr_t attr;
	if (pthread_attr_init(&attr) != 0) return -1;
	pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);
	int rc = pthread_create(th, &attr, fn, ctx);
	pthread_attr_destroy(&attr);
	return rc;
}

static size_t ring_size_for(size_t mtu, size_t mru)
{
	size_t a = PROD_MTU_MULT * mtu, b = CONS_MRU_MULT * mru;
	return a > b ? a : b;
}

int32_t block_connect_one2one(struct block *source, struct block *sink)
{
	if (!source || !sink || source->producer.type != PRODUCER_SINGLE || sink->consumer.type != CONSUMER_SINGLE ||
			source->producer.max_tu == 0) return 0;
	struct block_connection *c = hfdl_xcalloc(1, sizeof(*c));
	size_t cap = ring_size_for(source->producer.max_tu, sink->consumer.min_ru);
	size_t blk = hfdl_frontend_block_samples(sink);
	if (blk > 0) {
		/* The consumer is the GPU front end: page-locked storage, a whole number of its blocks long, so that every block it
		 * takes is one contiguous run it can DMA from in place: HFDL_GPU_PREFETCH_MAX + 5 blocks -- the block being pushed, up to
		 * HFDL_GPU_PREFETCH_MAX uploaded ahead of it (frontend.c), four for the producer to run ahead.
		 * If the producer is the library's own file input, the ring carries the file's RAW samples (cs16 / cu8 / cf32) and the
		 * conversion of src/input-helpers.c:33-78 happens on the device; any other producer gets the cf32 ring
		 * complex_samples_produce() expects. */
		size_t nblk = (cap + blk - 1) / blk;
		if (nblk < HFDL_GPU_PREFETCH_MAX + 5) nblk = HFDL_GPU_PREFETCH_MAX + 5;
		c->circ_buffer.buf = hfdl_ring_create_ex(nblk * blk, hfdl_file_input_raw_format(source), 1);
	} else {
		c->circ_buffer.buf = hfdl_ring_create(cap);
	}
	c->circ_buffer.cond = hfdl_xcalloc(1, sizeof(pthread_cond_t));
	c->circ_buffer.mutex = hfdl_xcalloc(1, sizeof(pthread_mutex_t));
	bool cond_ok = pthread_cond_init(c->circ_buffer.cond, NULL) == 0;
	bool mutex_ok = cond_ok && pthread_mutex_init(c->circ_buffer.mutex, NULL) == 0;
	if (!mutex_ok) {                 /* release the half-built connection */
		if (cond_ok) pthread_cond_destroy(c->circ_buffer.cond);
		hfdl_ring_destroy(c->circ_buffer.buf);
		free(c->circ_buffer.cond);
		free(c->circ_buffer.mutex);
		free(c);
		return 0;
	}
	source->producer.out = sink->consumer.in = c;
	return 1;
}

void block_disconnect_one2one(struct block *source, struct block *sink)
{
	if (!source || !sink || source->producer.out != sink->consumer.in || source->producer.out == NULL) return;
	struct block_connection *c = source->producer.out;
	hfdl_ring_destroy(c->circ_buffer.buf);
	pthread_cond_destroy(c->circ_buffer.cond);
	pthread_mutex_destroy(c->circ_buffer.mutex);
	free(c->circ_buffer.cond);
	free(c->circ_buffer.mutex);
	free(c);
	source->producer.out = sink->consumer.in = NULL;
}

int32_t block_connect_one2many(struct block *source, size_t sink_count, struct block *sinks[])
{
	if (!source || !sinks || source->producer.type != PRODUCER_MULTI || source->producer.max_tu == 0) return 0;
	for (size_t i = 0; i < sink_count; i++) if (sinks[i]->consumer.type != CONSUMER_MULTI) return 0;
	struct block_connection *c = hfdl_xcalloc(1, sizeof(*c));
	/* The reference hands the spectrum to the channels through shared_buffer.buf.  Here the spectrum never leaves
	 * HBM; the barriers are kept (sized sinks + 1) because they carry the start-up and shutdown hand-shake. */
	c->shared_buffer.buf = NULL;
	c->shared_buffer.data_ready = hfdl_xcalloc(1, sizeof(pthread_barrier_t));
	c->shared_buffer.consumers_ready = hfdl_xcalloc(1, sizeof(pthread_barrier_t));
	bool b1 = pthread_barrier_init(c->shared_buffer.data_ready, NULL, (unsigned)sink_count + 1) == 0;
	bool b2 = b1 && pthread_barrier_init(c->shared_buffer.consumers_ready, NULL, (unsigned)sink_count + 1) == 0;
	if (!b2) {
		if (b1) pthread_barrier_destroy(c->shared_buffer.data_ready);
		free(c->shared_buffer.data_ready);
		free(c->shared_buffer.consumers_ready);
		free(c);
		return 0;
	}
	source->producer.out = c;
	int32_t made = 0;
	for (size_t i = 0; i < sink_count; i++) { sinks[i]->consumer.in = c; made++; }
	return made;
}

void block_disconnect_one2many(struct block *source, size_t sink_count, struct block *sinks[])
{
	if (!source || !sinks || source->producer.out == NULL) return;
	struct block_connection *c = source->producer.out;
	for (size_t i = 0; i < sink_count; i++) if (sinks[i]->consumer.in == c) sinks[i]->consumer.in = NULL;
	pthread_barrier_destroy(c->shared_buffer.data_ready);
	pthread_barrier_destroy(c->shared_buffer.consumers_ready);
	free(c->shared_buffer.data_ready);
	free(c->shared_buffer.consumers_ready);
	free(c);
	source->producer.out = NULL;
}

void block_connection_one2one_shutdown(struct block_connection *connection)
{
	pthread_mutex_lock(connection->circ_buffer.mutex);
	connection->flags |= BLOCK_CONNECTION_SHUTDOWN;
	pthread_mutex_unlock(connection->circ_buffer.mutex);
	pthread_cond_signal(connection->circ_buffer.cond);
}

void block_connection_one2many_shutdown(struct block_connection *connection)
{
	connection->flags |= BLOCK_CONNECTION_SHUTDOWN;
	pthread_barrier_wait(connection->shared_buffer.data_ready);
}

bool block_connection_is_shutdown_signaled(struct block_connection *connection)
{
	return (connection->flags & BLOCK_CONNECTION_SHUTDOWN) != 0;
}

int32_t block_start(struct block *block)
{
	if (!block || !block->thread_routine) return 0;
	block->running = true;          /* set first: a short-lived thread may clear it before we return */
	if (hfdl_start_detached(&block->thread, block->thread_routine, block) != 0) {
		block->running = false;
		return 0;
	}
	return 1;
}

int32_t block_set_start(size_t block_cnt, struct block *block[])
{
	int32_t started = 0;
	for (size_t i = 0; i < block_cnt; i++) started += block_start(block[i]);
	return started;
}

bool block_is_running(struct block *block) { return block->running; }

bool block_set_is_any_running(size_t block_cnt, struct block *blocks[])
{
	for (size_t i = 0; i < block_cnt; i++) if (block_is_running(blocks[i])) return true;
	return false;
}
