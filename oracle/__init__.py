"""ORACLE package: CPU restatements of the reference algorithms used ONLY as a checker
(tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under misonet_amd/ imports it."""
