"""tools/bench_with.py -- bench.py with module switches of step_amd set first (A/B aid on the GPU box):

    python tools/bench_with.py backbone.POOL_WITH_POINTWISE=False driver.TUBE_KERNEL=False -- --config c3 --steps 60

Everything after `--` is bench.py's own command line."""
import ast
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    cut = argv.index("--") if "--" in argv else len(argv)
    for kv in argv[:cut]:
        path, val = kv.split("=", 1)
        mod, attr = path.rsplit(".", 1)
        m = importlib.import_module("step_amd." + mod)
        assert hasattr(m, attr), path
        setattr(m, attr, ast.literal_eval(val))
    import bench
    sys.argv = ["bench.py"] + argv[cut + 1:]
    bench.main()


if __name__ == "__main__":
    main()
