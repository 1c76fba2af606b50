"""Wider sweeps of the two CPU option-space fuzz tests than the collected seeds (not a pytest module):

    python tools/wide_cpu_fuzz.py [first_ref_seed n_ref_seeds first_list_seed n_list_seeds]

oracle vs the reference library over seeds 16..135, kernel-mode lists vs reference lists over seeds 12..111 (round 2's
ranges, the default); round 4 ran 136..335 and 112..311 (profiles/r04_cpu_fuzz_wide.txt)."""
import os
import sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_oracle_golden as t
from oracle_lib import load_oracle, load_ref
orc, ref = load_oracle(), load_ref()
a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [16, 120, 12, 100]
bad = 0
for seed in range(a[0], a[0] + a[1]):
    try:
        t.test_random_option_space_matches_reference.__wrapped__(orc, ref, seed) if hasattr(t.test_random_option_space_matches_reference, '__wrapped__') else t.test_random_option_space_matches_reference(orc, ref, seed)
    except AssertionError as e:
        bad += 1
        print("REF MISMATCH seed", seed, str(e)[:300], flush=True)
    except Exception as e:
        bad += 1
        print("REF ERROR seed", seed, repr(e)[:300], flush=True)
print("reference sweep over seeds %d..%d done, bad =" % (a[0], a[0] + a[1] - 1), bad, flush=True)
bad2 = 0
for seed in range(a[2], a[2] + a[3]):
    try:
        t.test_random_option_space_kernel_lists_equal_reference_lists(orc, seed)
    except AssertionError as e:
        bad2 += 1
        print("KERNEL-LIST MISMATCH seed", seed, str(e)[:300], flush=True)
    except Exception as e:
        bad2 += 1
        print("KERNEL-LIST ERROR seed", seed, repr(e)[:300], flush=True)
print("kernel-list sweep over seeds %d..%d done, bad =" % (a[2], a[2] + a[3] - 1), bad2, flush=True)
