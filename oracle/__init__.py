"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see oracle/kmpc_nlp.h)."""
