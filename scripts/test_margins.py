import sys, numpy as np, warnings
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
import tests.test_gpu_matrix as M, tests.test_gpu_callbacks as C
rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))
rows = []
for c in M.CASES:
    ft, fv, fp, fs = M._train("auto", *c); pt, pv, pp, ps = M._train("off", *c)
    rows.append(("matrix " + "-".join(str(x) for x in c if x not in (None, 1)), rel(ft, pt), rel(fv, pv), np.linalg.norm(fp - pp) / np.linalg.norm(pp)))
for name, (kind, change) in sorted(C.SCENARIOS.items()):
    ft, fv, fp, fs = C._train("auto", kind, change); pt, pv, pp, ps = C._train("off", kind, change)
    rows.append(("callback " + name, rel(ft, pt), rel(fv, pv), np.linalg.norm(fp - pp) / np.linalg.norm(pp)))
for r in sorted(rows, key=lambda r: -max(r[1:])):
    print(f"{r[0]:44s} train {r[1]:.1e} valid {r[2]:.1e} params {r[3]:.1e}")
