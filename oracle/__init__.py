"""ORACLE package: CPU restatements of the reference's hot path.

Test infrastructure only.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  Nothing under abnet3_amd/ imports this package.
"""
