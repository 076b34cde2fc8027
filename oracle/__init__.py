"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/smolmc_oracle.c)."""
