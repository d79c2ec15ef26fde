"""CPU oracle for the DeepFilterNet enhance() hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (deepfilternet_amd) never imports it and fails loudly when its HIP library is missing.

* df_oracle.c / libdf_oracle.py : C restatement of the reference's Rust DSP core (libDF + pyDF binding).
* dfnet_oracle.py               : torch-fp32 CPU restatement of DeepFilterNet3's forward and of enhance().
"""
