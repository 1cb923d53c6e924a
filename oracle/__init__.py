"""CPU oracle for the Conformer hybrid RNNT-CTC + CL training step.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package; the product (indic_cl_asr_amd) never does.
"""
