"""CPU oracle for the DC-VIC compress/decompress path.

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package (see DESIGN.md "Oracle").
"""
