"""Test-only empty stand-in (reference imports tensorboard.plugins.projector at import time)."""
