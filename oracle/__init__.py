"""CPU oracle for the ComputeMatches hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (monocularsfm_amd/) must never do so.

PARITY UNPINNED: the reference's arithmetic is a call into an un-vendored, un-pinned
OpenCV that is absent here, and the reference has no golden vectors for this path
(SURVEY.md section 4 / 8c).  `c_oracle` is the C restatement (msfm_oracle.c),
`np_oracle` an independent NumPy float32 restatement used to cross-check it.
"""
