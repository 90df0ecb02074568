"""Import stub (build container only)."""
