"""Measure the bf16-emulating oracle's own noise floors (its movement under a relative 1e-5 weight perturbation, per tensor; whole-network and
teacher-forced) for the full-width fixtures and cache them in tests/golden/<name>.floors.json -- CPU only, the oracle alone, no HIP library.

    python tests/make_floors.py [--reference-rounding] [name ...]        (--reference-rounding: the quad_emu=False entries of
    test_fullwidth_bf16_teacher_forced_vs_reference_graph_rounding; default: the fixtures of tests/test_fullwidth_gpu.py::test_fullwidth_bf16_vs_emulating_oracle)

tests/test_blocks_gpu.py::bf16_vs_emulating_oracle reads the cache (entries are keyed by the sha of oracle/restate.py: a stale entry is ignored and the
floors are measured inside the test instead), which saves one (D) or two (G) of its oracle passes at full width."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import test_blocks_gpu as TB  # noqa: E402

if __name__ == "__main__":
    ref_rounding = "--reference-rounding" in sys.argv[1:]
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["biggan128w", "sngan32w", "wgangp128w", "bigdeep128w"]
    for name in names:
        path = TB._floors_path(name)
        tab = json.load(open(path)) if os.path.exists(path) else {}
        for which in ("D", "G"):
            t0 = time.time()
            r = TB.bf16_vs_emulating_oracle(name, which, dev=torch.device("cpu"), floors_only=True, quad_emu=not ref_rounding)
            tab[r["key"]] = r["entry"]
            e = r["entry"]
            print(f"{name} {which}: whole-network floor {e['whole_floor']:.3e}, {len(e['fl'])} tensors, worst teacher-forced floor "
                  f"{max(e['tfl'].values()) if e['tfl'] else 0.0:.3e} ({time.time() - t0:.0f} s)", flush=True)
        json.dump(tab, open(path, "w"), indent=0, sort_keys=True)
