"""Host-side housekeeping of the step loop (no effect on what is computed).

`freeze_gc()`: a train step of this package creates a few thousand short-lived Python objects (autograd nodes, ctypes
argument structs, tensor views).  CPython's cyclic collector then runs a full (generation-2) collection every few dozen
steps, and with ~1e6 long-lived objects in the process (torch, the model, the optimiser) that pass takes ~30 ms -- more than
the host's lead over the GPU on the B=16 workloads: measured on MI355X (SCD, B=16, T=5) one 10-step window in three ran at
25.8 instead of 22.4 ms per step, i.e. 677 instead of 714 img/s for a 30-step measurement; with the collector disabled every
window is 22.4.  `gc.freeze()` after the first steps moves everything alive at that point into the permanent generation:
later full collections only look at what the steps created since, and stay under a millisecond.  The training-script
mirrors call it once per run after their first iterations; `bench.py` calls it before the timed region."""
import gc


def freeze_gc():
    gc.collect()
    gc.freeze()
