"""Measurement apparatus (store-pattern / sincos / whole-tensor probes) -- not on the product path.
See include/dcs_probes.h."""
