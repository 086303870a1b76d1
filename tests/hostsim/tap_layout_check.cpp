// tap_layout_check.cpp -- host-side check of kernels.h tap_index_f (TAPL_OCTET): the filter taps are stored in the operand order of
// v_mfma_f32_16x16x4_f32.  Compiled and run by tests/test_host_logic_cpu.py (no GPU): prints "ok" or the first violation.
//   * a bijection of (row, channel, bin, comp) onto [0, p * nch_pad * m * 2)
//   * a 1 KiB chunk = 4 alias rows x 8 channels x 4 bins with lane = 16 (row % 4) + 2 (c % 8) + comp and register = bin % 4, i.e.
//     operand A of the instruction for bin j is register j % 4 of the chunk of bin quad j / 4: A[i = 2 (c % 8) + comp][k = row % 4]
//   * the four bin quads of a 16-bin tile are consecutive KiB; the octets of a quad of rows follow each other; Im sits one lane on
#include <cstdio>
#include <vector>
#include "kernels.h"

using namespace hfdl;

int main()
{
	const int m = 64, p = 16, nch_pad = 24;
	const size_t rs_f = (size_t)nch_pad * m * 2, total = (size_t)p * rs_f;
	std::vector<char> seen(total, 0);
	for (int row = 0; row < p; row++)
		for (int c = 0; c < nch_pad; c++)
			for (int j = 0; j < m; j++)
				for (int comp = 0; comp < 2; comp++) {
					const size_t at = tap_index_f(TAPL_OCTET, m, rs_f, c, row, j, comp);
					if (at >= total || seen[at]) { printf("not a bijection at row %d c %d j %d comp %d -> %zu\n", row, c, j, comp, at); return 1; }
					seen[at] = 1;
					const size_t chunk = at / 256, in = at % 256, lane = in / 4, reg = in % 4;
					if (lane != (size_t)(16 * (row & 3) + 2 * (c & 7) + comp) || reg != (size_t)(j & 3)) { printf("operand order broken at row %d c %d j %d\n", row, c, j); return 1; }
					const size_t want_chunk = ((size_t)(row >> 2) * (nch_pad / 8) + (c >> 3)) * (m / 4) + (j >> 2);
					if (chunk != want_chunk) { printf("chunk order broken at row %d c %d j %d: %zu != %zu\n", row, c, j, chunk, want_chunk); return 1; }
					if (comp == 1 && at != tap_index_f(TAPL_OCTET, m, rs_f, c, row, j, 0) + 4) { printf("Im is not one lane on\n"); return 1; }
				}
	// TAPL_PLAIN: rows of m cf32 per channel
	for (int row = 0; row < 3; row++)
		for (int c = 0; c < 5; c++)
			for (int j = 0; j < m; j++)
				if (tap_index_f(TAPL_PLAIN, m, rs_f, c, row, j, 1) != (size_t)row * rs_f + ((size_t)c * m + j) * 2 + 1) { printf("plain layout broken\n"); return 1; }
	printf("ok\n");
	return 0;
}
