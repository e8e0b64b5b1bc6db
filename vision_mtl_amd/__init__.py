"""MI355X (gfx950) implementation of the vision_mtl training / inference step path.

`_lib` binds the HIP kernels behind the C ABI of include/vmtl.h, `ops` wraps them as autograd functions,
`layers` / `models` / `lit_module` mirror the reference's module surface, `dp` is the one-process-per-GPU
data-parallel layer (flat gradient arena, one RCCL all-reduce per step)."""
