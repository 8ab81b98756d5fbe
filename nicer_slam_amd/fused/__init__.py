"""Fused gfx950 engine of the render core: weight packing + thin wrappers over Section 2 of the C ABI."""
