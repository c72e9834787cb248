"""CPU oracle — test infrastructure only (see oracle/minarrow_oracle.c). Never imported by minarrow_amd."""
