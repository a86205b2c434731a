"""Layers for use in models (graph_weather/models/layers of the reference)."""
