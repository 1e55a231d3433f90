"""Oracle = CPU restatement of the reference algorithm. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
path (aether_amd/, aether/, scripts/) never does.  Status: PARITY UNPINNED — the arithmetic of this path lives
in the third-party `diffusers` package (>=0.32.2, /root/reference/requirements.txt:4) which is neither vendored
under /root/reference nor installable in the build container, and the reference ships no golden vectors.
Pinned exceptions (round 2): oracle/rope.py and the orchestration restated by oracle/pipeline.py equal the outputs of the
reference's OWN pipeline module run in the build container (tools/make_golden.py -> tests/golden/pipeline.npz).
"""
