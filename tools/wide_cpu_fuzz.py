"""Wider sweeps of the two CPU option-space fuzz tests than the collected seeds (not a pytest module):

    cd tests && python wide_cpu_fuzz.py

oracle vs the reference library over seeds 16..135, kernel-mode lists vs reference lists over seeds 12..111."""
import os
import sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_oracle_golden as t
from oracle_lib import load_oracle, load_ref
orc, ref = load_oracle(), load_ref()
bad = 0
for seed in range(16, 136):
    try:
        t.test_random_option_space_matches_reference.__wrapped__(orc, ref, seed) if hasattr(t.test_random_option_space_matches_reference, '__wrapped__') else t.test_random_option_space_matches_reference(orc, ref, seed)
    except AssertionError as e:
        bad += 1
        print("REF MISMATCH seed", seed, str(e)[:300], flush=True)
    except Exception as e:
        bad += 1
        print("REF ERROR seed", seed, repr(e)[:300], flush=True)
print("reference sweep done, bad =", bad, flush=True)
bad2 = 0
for seed in range(12, 112):
    try:
        t.test_random_option_space_kernel_lists_equal_reference_lists(orc, seed)
    except AssertionError as e:
        bad2 += 1
        print("KERNEL-LIST MISMATCH seed", seed, str(e)[:300], flush=True)
    except Exception as e:
        bad2 += 1
        print("KERNEL-LIST ERROR seed", seed, repr(e)[:300], flush=True)
print("kernel-list sweep done, bad =", bad2, flush=True)
