"""CPU oracles for the VB E-step hot path - TEST INFRASTRUCTURE ONLY.

vb_numpy  numpy/scipy restatement in the reference's operation order
c_oracle  ctypes wrapper of vb_oracle.c (plain C restatement)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; pylda_amd never does.
"""
