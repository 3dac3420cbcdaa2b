"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the PTv3 / SpUNet hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package; nothing under
pointcept_amd/ imports it, and the product ops raise if the HIP library is missing instead of
falling back to anything here.

Parity pinning status ("Oracle"):
  * serialization codes, orders, pad maps, pooling maps : pinned against the reference's own
    Python code imported in the authoring container (tests/golden/make_golden.py -> fixtures).
  * spconv / flash_attn / torch_scatter arithmetic      : the modules are third-party and absent
    from /root/reference ("parity unpinned" against the real libraries); restated from their
    documented semantics and cross-checked against dense F.conv3d / SDPA / brute-force loops.
"""
