"""Per-tensor table of the teacher-forced bf16 comparison of a full-width fixture (tests/test_blocks_gpu.py bf16_vs_emulating_oracle) with the kernels run on the
CPU interpreter (tests/hipemu): every block output, block-input gradient and weight gradient of the network against the bf16-emulating oracle, next to the
oracle's own noise floor for that tensor (tests/golden/<name>.floors.json). Answers VERDICT r4 item 6 ("print the per-block table, find the first block above
2e-2") without GPU time: the interpreter executes the kernel sources lane by lane (it does not know the summation order inside an MFMA, nothing else differs).
usage: python tools/teacher_table.py bigdeep128w G > profiles/<name>.txt        (TEST INFRASTRUCTURE; minutes per network)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


def main():
    import fullemu
    import studiogan_amd  # noqa: F401
    import test_blocks_gpu as TB
    name, which = sys.argv[1], sys.argv[2]
    torch.set_num_threads(1)
    t = time.time()
    with fullemu.Installed(dma_late=1, greedy=1, seed=1):
        try:
            e, fl = TB.bf16_vs_emulating_oracle(name, which, dev=torch.device("cpu"))
            print(f"# {name} {which}: whole-network gradient error {e:.3e}, its floor {fl:.3e}")
        except AssertionError as ex:
            print("# comparison reported mismatches:", str(ex)[:2000])
    print(f"# {time.time() - t:.0f} s on the interpreter")


if __name__ == "__main__":
    main()
