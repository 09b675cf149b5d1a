"""One fresh process of tests/test_upload_order.py: ganon-build of the reference's ten 80-mers, then -- as this process's
first GPU work -- load_ibf -> submit -> fetch -> dense tap, checked like GanonBuild.test.cpp's validate_elements."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

if len(sys.argv) > 4 and sys.argv[4] == "torch":  # PyTorch's bundled HIP runtime in the process first, as under pytest and in bench.py
    import torch  # noqa: E402
    torch.cuda.is_available()
import ganon_amd  # noqa: E402
import test_build_gpu as t  # noqa: E402

d, k, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
inp, _, names = t.write_inputs(d, t.SEQS)
out, _ = t.run_build(d, inp, k=k, w=w)
t.check_filter(ganon_amd, out, t.SEQS, names, k, w, 4, 0.05, 0)
print("ok")
