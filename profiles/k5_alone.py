#!/usr/bin/env python3
"""Viterbi cycle probe (debug build with -DHFDL_VIT_DEBUG prints forward / chainback cycles of workgroup 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dumphfdl_amd as hf
rng = np.random.default_rng(0)
for nframes in (8, 2048, 2048):
    hf.viterbi27(rng.integers(0, 256, (nframes, 2 * 7560), dtype=np.uint8), 7560)
