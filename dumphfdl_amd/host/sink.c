/* sink.c -- objects that cross the downstream boundary (src/util.c:100-105, src/pdu.c:81-85) and a weak default
 * pdu_decoder_queue_push() so tools without the protocol parsers still link. */
#include <stdio.h>
#include <stdlib.h>
#include "hfdl_host.h"
#include "host_internal.h"

struct octet_string *octet_string_new(void *buf, size_t len)
{
	struct octet_string *o = hfdl_xcalloc(1, sizeof(*o));
	o->buf = buf;
	o->len = len;
	return o;
}

void octet_string_destroy(struct octet_string *o)
{
	if (o == NULL) return;
	free(o->buf);
	free(o);
}

static struct metadata *meta_copy(struct metadata const *m)
{
	struct hfdl_pdu_metadata *c = hfdl_xcalloc(1, sizeof(*c));
	*c = *container_of(m, struct hfdl_pdu_metadata, metadata);
	return &c->metadata;
}

static void meta_destroy(struct metadata *m)
{
	if (m) free(container_of(m, struct hfdl_pdu_metadata, metadata));
}

static struct metadata_vtable pdu_meta_vtable = { meta_copy, meta_destroy };

struct metadata *hfdl_pdu_metadata_create(void)
{
	struct hfdl_pdu_metadata *m = hfdl_xcalloc(1, sizeof(*m));
	m->metadata.vtable = &pdu_meta_vtable;
	return &m->metadata;
}

static pthread_mutex_t print_lock = PTHREAD_MUTEX_INITIALIZER;

__attribute__((weak)) void pdu_decoder_queue_push(struct metadata *metadata, struct octet_string *pdu, uint32_t flags)
{
	(void)flags;
	if (metadata == NULL || pdu == NULL) return;
	struct hfdl_pdu_metadata *hm = container_of(metadata, struct hfdl_pdu_metadata, metadata);
	pthread_mutex_lock(&print_lock);
	printf("PDU freq=%d bit_rate=%d slot=%c freq_err=%.2f rssi=%.1f nf=%.1f ts=%ld.%06ld len=%zu ",
			hm->freq, hm->bit_rate, hm->slot, hm->freq_err_hz, hm->rssi, hm->noise_floor,
			(long)metadata->rx_timestamp.tv_sec, (long)metadata->rx_timestamp.tv_usec, pdu->len);
	for (size_t i = 0; i < pdu->len; i++) printf("%02x", pdu->buf[i]);
	printf("\n");
	fflush(stdout);
	pthread_mutex_unlock(&print_lock);
	octet_string_destroy(pdu);
	metadata->vtable->destroy(metadata);
}
