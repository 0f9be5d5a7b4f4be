"""TEST INFRASTRUCTURE ONLY. See oracle/bigint_oracle.py and oracle/ark_cpu.cpp headers."""
