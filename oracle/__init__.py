"""ORACLE — test infrastructure only (see oracle/hotpath_ref.py).  Not part of the product; the product never
imports this package."""
