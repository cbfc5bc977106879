"""CPU oracle (test infrastructure only). See oracle/qp_oracle.c."""
