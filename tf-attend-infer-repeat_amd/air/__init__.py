"""Host-side mirror of the reference's ``air`` package (air/air_model.py,
air/transformer.py) on top of libair_hip.so."""
