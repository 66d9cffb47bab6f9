"""CPU oracle for the fabstir-vectordb distance-computation hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.
See oracle/oracle.cpp for the restatement and its reference citations.
"""
from .oracle import *  # noqa: F401,F403
