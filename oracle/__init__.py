"""Oracle = CPU restatement of the reference algorithm. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
path (aether_amd/, aether/, scripts/) never does.  Status: PARITY UNPINNED — the arithmetic of this path lives
in the third-party `diffusers` package (>=0.32.2, /root/reference/requirements.txt:4) which is neither vendored
under /root/reference nor installable in the build container, and the reference ships no golden vectors.
Pinned exceptions (round 2): oracle/rope.py and the orchestration restated by oracle/pipeline.py equal the outputs of the
reference's OWN pipeline module run in the build container (tools/make_golden.py -> tests/golden/pipeline.npz).
Round 3: tools/make_fullsize_golden.py also runs this package OFFLINE at the full depth and geometry (42 blocks, S = 15 076, whole-clip VAE,
4-step trajectory) and stores its outputs as fixtures (tests/golden/fullsize_*.npz) for tests/test_fullsize_parity_gpu.py — outputs of the
ORACLE, i.e. still un-pinned against diffusers; tools/fullsize_cases.py and tools/make_fullsize_golden.py are test infrastructure like tests/.
"""
