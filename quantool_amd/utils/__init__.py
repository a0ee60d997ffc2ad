"""Host-side helpers around the calibration path (no device code)."""
