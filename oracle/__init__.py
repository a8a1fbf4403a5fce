"""CPU oracle for libptops — TEST INFRASTRUCTURE ONLY (see oracle/ptops_ref.c header)."""
