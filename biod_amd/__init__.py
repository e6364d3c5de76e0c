"""biod_amd -- MI355X-native BAM decode -> CIGAR walk -> pileup (BioD's hot path) behind a C ABI.

`biod_amd.libbiodgpu.so` (hand-written HIP for gfx950) is the product; this package is the thin
host-side mirror of BioD's interface used by the tests, the benchmark and Python callers."""
__all__ = ["bamio", "pileup", "synth"]
