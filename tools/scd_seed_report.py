import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_model_gpu as t
for w in (23, 26, 27, 16, 64, 66, 61):
    errs = t._conditioned_case("scd", w)
    worst, med, n_over = t._summ("scd", w, errs)
    print("SEED", w, "worst %.2e med %.2e over %d/%d" % (worst, med, n_over, len(errs)), flush=True)
