"""Size arithmetic of the P1 patterns and the algorithmic byte
count of one CSR SpMV used for the roofline figure."""


def nnz_p1_box(n: int) -> int:
    """15-point pattern of P1 tetrahedra (6 per cube, shared main diagonal) on n^3 vertices."""
    return n ** 3 + 6 * n * n * (n - 1) + 6 * n * (n - 1) ** 2 + 2 * (n - 1) ** 3


def nnz_p1_rect(n: int) -> int:
    return n * n + 4 * n * (n - 1) + 2 * (n - 1) ** 2


def spmv_bytes(n: int, nnz: int) -> int:
    """values + column ids, row_ptr, one read of x, one write of y"""
    return nnz * (8 + 4) + n * (4 + 8 + 8)
