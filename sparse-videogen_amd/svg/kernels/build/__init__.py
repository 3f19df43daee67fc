"""Module path of the reference's built CUDA extension (`svg/kernels/build/_kernels*.so`, loaded through
`sys.path.append('svg/kernels/build/')` + `import _kernels`, ref: svg/models/hyvideo/attention.py:159-160).
Here `_kernels` is a Python module over libsvgattn's C ABI — nothing is compiled into this directory."""
