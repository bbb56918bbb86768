"""One of bench.py's extra workloads alone (for a kernel trace of it):
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o kt --output-format csv -- python <repo>/tools/extra_run.py bigdeep128_bs256_bf16"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    print(json.dumps(bench.run_extra(name, torch.device("cuda:0"), steps=steps, warmup=1)))
