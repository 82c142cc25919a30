"""Name-only stub of `timm` (not installed in the build container). Test infrastructure."""
