"""ref: svg/models/utils.py — small helpers shared by the model packages."""


def visualize_sparse_bsr(indptr, indices, block_size, max_rows: int = 40) -> str:
    """ASCII picture of a BSR mask (debug aid of the reference's flashinfer back-end)."""
    rows = len(indptr) - 1
    ncols = (max(indices) + 1) if len(indices) else 0
    lines = []
    for r in range(min(rows, max_rows)):
        cols = set(int(c) for c in indices[int(indptr[r]):int(indptr[r + 1])])
        lines.append("".join("#" if c in cols else "." for c in range(ncols)))
    return "\n".join(lines)
