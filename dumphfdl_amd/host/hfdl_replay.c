/* hfdl_replay.c -- minimal host program over libhfdl_host.so: the wiring of dumphfdl's main() (src/main.c:687-802) for a raw
 * I/Q file, with the protocol parsers replaced by the library's printing pdu_decoder_queue_push().
 *
 *   hfdl_replay --iq-file FILE --sample-rate HZ --sample-format CF32|CS16|CU8 --centerfreq KHZ [--device N]
 *               [--statsd-print] [--noise-floor-stats-interval S] [--shard R/W] [--bench] [--loop N] FREQ_KHZ...
 *
 * --shard R/W   single-stream multi-GPU mode: this process decodes channels R, R+W, R+2W, ... of the list on
 *               its --device; W processes fed the same file cover all channels, no communication between them.
 * --bench       PDUs are counted instead of printed and one JSON line reports Msamples/s of the whole host path
 *               (file in page cache -> input block -> ring -> GPU front end -> pdu_decoder_queue_push), create time excluded.
 * --loop N      replay the file N times back to back (bench runs longer than the file).
 *
 * --statsd-print stands in for dumphfdl's src/statsd.c: the strong statsd_* hooks below print one "STATSD" line per
 * counter total at exit and one per noise-floor gauge as it arrives (metric names as in doc/STATSD_METRICS.md).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hfdl_host.h"

static int statsd_print, bench_mode;
static unsigned long long bench_pdus, bench_octets;

/* In --bench mode this strong definition replaces the library's printing default: count and free (ownership is ours,
 * src/pdu.c:37-43, 171-172).  Otherwise print the same line as the library's default so the output stays comparable. */
void pdu_decoder_queue_push(struct metadata *metadata, struct octet_string *pdu, uint32_t flags)
{
	(void)flags;
	if (metadata == NULL || pdu == NULL) return;
	if (bench_mode) {
		bench_pdus++;
		bench_octets += pdu->len;
	} else {
		struct hfdl_pdu_metadata *hm = (struct hfdl_pdu_metadata *)metadata;      /* metadata is its first member */
		printf("PDU freq=%d bit_rate=%d slot=%c freq_err=%.2f rssi=%.1f nf=%.1f ts=%ld.%06ld len=%zu ",
				hm->freq, hm->bit_rate, hm->slot, hm->freq_err_hz, hm->rssi, hm->noise_floor,
				(long)metadata->rx_timestamp.tv_sec, (long)metadata->rx_timestamp.tv_usec, pdu->len);
		for (size_t i = 0; i < pdu->len; i++) printf("%02x", pdu->buf[i]);
		printf("\n");
		fflush(stdout);
	}
	octet_string_destroy(pdu);
	metadata->vtable->destroy(metadata);
}

static struct tally { int32_t freq; unsigned a2, m1, m1_missing; } tallies[4096];
static int ntallies;

static struct tally *tally_of(int32_t freq)
{
	for (int i = 0; i < ntallies; i++) if (tallies[i].freq == freq) return &tallies[i];
	if (ntallies == 4096) return NULL;
	tallies[ntallies].freq = freq;
	return &tallies[ntallies++];
}

/* called from the front-end thread only */
void statsd_counter_per_channel_increment(int32_t freq, char *counter)
{
	struct tally *t = tally_of(freq);
	if (t == NULL) return;
	if (!strcmp(counter, "demod.preamble.A2_found")) t->a2++;
	else if (!strcmp(counter, "demod.preamble.M1_found")) t->m1++;
	else if (!strcmp(counter, "demod.preamble.errors.M1_not_found")) t->m1_missing++;
}

void statsd_gauge_per_channel_set(int32_t freq, char *gauge, size_t value)
{
	if (statsd_print) fprintf(stderr, "STATSD gauge %d.%s %zu\n", freq, gauge, value);
}

int main(int argc, char **argv)
{
	struct input_cfg *cfg = input_cfg_create();
	cfg->type = INPUT_TYPE_FILE;
	double centerfreq_khz = -1;
	int32_t freqs[4096];
	int nfreq = 0, shard_rank = 0, shard_world = 1;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "--iq-file") && i + 1 < argc) cfg->source = argv[++i];
		else if (!strcmp(argv[i], "--sample-rate") && i + 1 < argc) cfg->sample_rate = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--sample-format") && i + 1 < argc) cfg->sfmt = sample_format_from_string(argv[++i]);
		else if (!strcmp(argv[i], "--centerfreq") && i + 1 < argc) centerfreq_khz = atof(argv[++i]);
		else if (!strcmp(argv[i], "--read-buffer-size") && i + 1 < argc) cfg->read_buffer_size = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--device") && i + 1 < argc) hfdl_frontend_set_device(atoi(argv[++i]));
		else if (!strcmp(argv[i], "--statsd-print")) statsd_print = 1;
		else if (!strcmp(argv[i], "--bench")) bench_mode = 1;
		else if (!strcmp(argv[i], "--loop") && i + 1 < argc) hfdl_file_input_set_loops(atoi(argv[++i]));
		else if (!strcmp(argv[i], "--shard") && i + 1 < argc) {
			if (sscanf(argv[++i], "%d/%d", &shard_rank, &shard_world) != 2 || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) {
				fprintf(stderr, "--shard wants RANK/WORLD with 0 <= RANK < WORLD\n");
				return 1;
			}
		}
		else if (!strcmp(argv[i], "--noise-floor-stats-interval") && i + 1 < argc) hfdl_nf_stats_set_interval(atoi(argv[++i]));
		else {
			/* anything that reads as a number is a channel frequency in kHz (kHz -> Hz, src/main.c:197-212); synthetic plans
			 * may carry negative ones */
			char *end = NULL;
			double khz = strtod(argv[i], &end);
			if (end == argv[i] || *end != '\0' || nfreq >= 4096) { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
			freqs[nfreq++] = (int32_t)lround(1e3 * khz);
		}
	}
	if (!cfg->source || cfg->sample_rate < HFDL_SYMBOL_RATE * SPS || cfg->sfmt == SFMT_UNDEF || nfreq == 0) {
		fprintf(stderr, "usage: %s --iq-file F --sample-rate HZ --sample-format FMT [--centerfreq KHZ] freq_khz...\n", argv[0]);
		return 1;
	}
	if (centerfreq_khz < 0) {        /* midpoint of the outermost channels, src/main.c:228-239 */
		int32_t lo = freqs[0], hi = freqs[0];
		for (int i = 1; i < nfreq; i++) { if (freqs[i] < lo) lo = freqs[i]; if (freqs[i] > hi) hi = freqs[i]; }
		cfg->centerfreq = lo + (hi - lo) / 2;
	} else {
		cfg->centerfreq = (int32_t)(1e3 * centerfreq_khz);
	}
	if (shard_world > 1) {           /* round-robin channel partition; the centre frequency above is the whole list's, shared by all shards */
		int kept = 0;
		for (int i = 0; i < nfreq; i++) if (i % shard_world == shard_rank) freqs[kept++] = freqs[i];
		nfreq = kept;
		if (nfreq == 0) { fprintf(stderr, "shard %d/%d holds no channel\n", shard_rank, shard_world); return 1; }
	}
	struct block *input = input_create(cfg);
	if (input == NULL || input_init(input) < 0) { fprintf(stderr, "Unable to initialize input\n"); return 1; }
	int32_t decimation = compute_fft_decimation_rate(cfg->sample_rate, HFDL_SYMBOL_RATE * SPS);
	float tbw = compute_filter_relative_transition_bw(cfg->sample_rate, HFDL_CHANNEL_TRANSITION_BW_HZ);
	struct block *fft = fft_create(decimation, tbw);
	if (fft == NULL) return 1;
	hfdl_init_globals();
	struct block **channels = calloc((size_t)nfreq, sizeof(*channels));
	for (int i = 0; i < nfreq; i++) {
		channels[i] = hfdl_channel_create(cfg->sample_rate, decimation, tbw, cfg->centerfreq, freqs[i]);
		if (channels[i] == NULL) { fprintf(stderr, "Failed to initialize channel %d\n", freqs[i]); return 1; }
	}
	if (block_connect_one2one(input, fft) != 1 || block_connect_one2many(fft, (size_t)nfreq, channels) != nfreq) return 1;
	/* start order: channels, fft, input (src/main.c:770-774) */
	if (block_set_start((size_t)nfreq, channels) != nfreq || block_start(fft) != 1 || block_start(input) != 1) return 1;
	if (hfdl_nf_stats_thread_start(channels, nfreq) < 0) return 1;             /* src/main.c:777-783 */
	while (block_is_running(input) || block_is_running(fft) || block_set_is_any_running((size_t)nfreq, channels)) usleep(20000);
	hfdl_print_summary();
	if (bench_mode) {
		struct hfdl_run_stats rs;
		hfdl_frontend_run_stats(&rs);
		printf("{\"tool\": \"hfdl_replay --bench\", \"value\": %.3f, \"unit\": \"Msamples/s\", \"samples\": %llu, \"blocks\": %llu, \"seconds\": %.6f, "
				"\"pdus\": %llu, \"pdu_octets\": %llu, \"channels\": %d, \"block_samples\": %d, \"bytes_per_sample_over_pcie\": %d, \"zero_copy_ring\": %s, \"thread_s\": {\"wait_input\": %.4f, \"enqueue\": %.4f, \"collect\": %.4f, \"release\": %.4f, \"grace\": %.4f}, \"pipeline_drains\": %llu, "
				"\"lpdu_walk_on_device\": {\"mpdus\": %llu, \"lpdus\": %llu, \"good_fcs\": %llu, \"bad_fcs\": %llu}, \"shard\": \"%d/%d\", \"path\": \"file (page cache) -> file input (parallel pread into the page-locked ring) -> GPU front-end block -> pdu_decoder_queue_push\"}\n",
				rs.seconds > 0 ? (double)rs.samples / rs.seconds / 1e6 : 0.0, (unsigned long long)rs.samples, (unsigned long long)rs.blocks, rs.seconds,
				bench_pdus, bench_octets, rs.channels, rs.block_samples, rs.bytes_per_sample, rs.zero_copy ? "true" : "false", rs.wait_input_s, rs.push_s, rs.collect_s, rs.release_s, rs.grace_s, (unsigned long long)rs.drains,
				(unsigned long long)rs.mpdus_walked, (unsigned long long)rs.lpdus_processed, (unsigned long long)rs.lpdus_good, (unsigned long long)rs.lpdus_bad_fcs, shard_rank, shard_world);
		fflush(stdout);
	}
	if (statsd_print)
		for (int i = 0; i < ntallies; i++)
			fprintf(stderr, "STATSD counter %d A2_found=%u M1_found=%u M1_not_found=%u\n", tallies[i].freq, tallies[i].a2, tallies[i].m1, tallies[i].m1_missing);
	block_disconnect_one2many(fft, (size_t)nfreq, channels);
	block_disconnect_one2one(input, fft);
	for (int i = 0; i < nfreq; i++) hfdl_channel_destroy(channels[i]);
	fft_destroy(fft);
	input_destroy(input);
	input_cfg_destroy(cfg);
	free(channels);
	return 0;
}
