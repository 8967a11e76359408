"""CPU oracle for the POCO hot path - TEST INFRASTRUCTURE ONLY (see oracle/poco_ref.py header)."""
