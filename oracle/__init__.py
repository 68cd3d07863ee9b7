"""CPU oracle (test infrastructure only -- see oracle/cvodes_oracle.c header)."""
