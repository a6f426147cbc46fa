"""compat shim package (see compat/README.md)."""
