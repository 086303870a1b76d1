"""dumphfdl_amd -- MI355X (gfx950) native multichannel HFDL front end.

Drop-in for dumphfdl's hot path (fastddc channelizer + per-channel HFDL demod/FEC) behind a C ABI
(include/hfdl_gpu.h -> libhfdl_gpu.so).  This package is the thin Python mirror used by the tests and
bench.py; the host program proper stays C (dumphfdl_amd/host, include/hfdl_host.h).

There is no CPU path: every entry point fails loudly when libhfdl_gpu.so or a gfx950 device is missing.
"""
from .frontend import Frontend, GpuError, lib_path, load, fft_forward, viterbi27, burst_decode, device_count, plan_geometry, host_alloc, host_free, nco_decimate, crc16_ccitt, pdu_triage, lpdu_walk, psk_slice  # noqa: F401
