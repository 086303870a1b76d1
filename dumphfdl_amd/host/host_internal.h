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

#define container_of(ptr, type, member) ((type *)((char *)(ptr) - offsetof(type, member)))
