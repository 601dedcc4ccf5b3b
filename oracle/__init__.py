"""CPU oracle package (TEST INFRASTRUCTURE ONLY — see oracle/oracle.c and oracle/pyref.py headers)."""
