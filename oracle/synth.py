"""ORACLE — test infrastructure only.

The deterministic synthetic-data generator lives in `change3d_amd/synthetic.py` (a neutral module:
benchmarks and scripts need seeded tensors without importing the oracle); re-exported here so
oracle-side code and tests keep one definition."""
from change3d_amd.synthetic import (make_args, make_cc_args, synth_batch, synth_captions, synth_scd_labels, synth_state_dict,  # noqa: F401
                                    synth_tensor)
