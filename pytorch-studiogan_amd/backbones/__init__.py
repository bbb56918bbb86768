"""Backbone modules importable by name like the reference's `models.<backbone>` (reference src/models/model.py:22)."""
