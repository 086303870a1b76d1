// h2d_streams.hip -- does the host -> device rate of hipMemcpyAsync depend on WHICH stream (hardware queue / copy engine) carries it?
// 8 non-blocking streams created one after the other, the same 7.3 MB page-locked block copied 6 times on each (events on that stream).
// hipcc --offload-arch=gfx950 -O2 h2d_streams.hip -o h2d_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
	const size_t bytes = 917504 * 8;
	void *h[2], *d[2];
	for (int i = 0; i < 2; i++) { CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault)); memset(h[i], 1, bytes); CK(hipMalloc(&d[i], bytes)); }
	std::vector<hipStream_t> st(8);
	for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int pass = 0; pass < 2; pass++)
		for (size_t i = 0; i < st.size(); i++) {
			CK(hipMemcpyAsync(d[0], h[0], bytes, hipMemcpyHostToDevice, st[i]));
			CK(hipStreamSynchronize(st[i]));
			float best = 1e9f, sum = 0;
			for (int r = 0; r < 6; r++) {
				CK(hipEventRecord(e0, st[i]));
				CK(hipMemcpyAsync(d[r & 1], h[r & 1], bytes, hipMemcpyHostToDevice, st[i]));
				CK(hipEventRecord(e1, st[i]));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best; sum += ms;
			}
			printf("pass %d stream %zu: best %.1f GB/s, mean %.1f GB/s\n", pass, i, bytes / (best * 1e-3) / 1e9, bytes / (sum / 6 * 1e-3) / 1e9);
		}
	return 0;
}
