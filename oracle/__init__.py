"""CPU oracle for the U-ViT flow-matching hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline.  The product path
(``uspace_amd``) never imports it and fails loudly without its HIP library.
"""
