"""CPU oracle package (test infrastructure only -- see oracle/mjstep.c)."""
