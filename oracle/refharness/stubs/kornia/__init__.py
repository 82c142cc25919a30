"""Name-only stub of `kornia` (not installed). Test infrastructure."""
