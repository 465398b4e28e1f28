"""tools/bench_with.py -- bench.py with module switches of step_amd set first (A/B aid on the GPU box):

    python tools/bench_with.py backbone.POOL_WITH_POINTWISE=False driver.TUBE_KERNEL=False -- --config c3 --steps 60
    python tools/bench_with.py lib=wgc3 -- --config c4 --dtype bf16          (an experiment build, `make -C step_amd/csrc EXP=wgc3 EXPFLAGS=...`)

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
        if path == "lib":                                   # lib=NAME: an experiment build tools/libstep_amd_NAME.so instead of the product library
            import ctypes
            from step_amd import _capi, _lib
            L_ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_%s.so" % val))
            _capi.declare(L_, strict=False)
            _lib._LIB = L_
            continue
        mod, attr = path.rsplit(".", 1)
        m = importlib.import_module("step_amd." + mod)
        assert hasattr(m, attr), path
        setattr(m, attr, ast.literal_eval(val))
    import bench
    sys.argv = ["bench.py"] + argv[cut + 1:]
    bench.main()


if __name__ == "__main__":
    main()
