"""Seeded synthetic scenes, networks and noise for tests, golden generation and bench.py.
Test/measurement infrastructure: nothing under pixel-nerf_amd/ (the product) imports this package."""
