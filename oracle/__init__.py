"""CPU oracle for the V2V hot path -- test infrastructure only (see oracle/v2v_oracle.py header)."""
