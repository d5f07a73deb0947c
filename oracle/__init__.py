"""CPU ORACLE package - test infrastructure only (see otpose_oracle.py header)."""
