# import stub (not reference code, not product code): lets oracle/gen_golden.py import the
# reference's modules in the build container, where absl is not installed.
