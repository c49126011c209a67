"""starkperp - Python host side of the MI355X hot path for the StarkEx-Perpetual crypto builtins.

`starkperp.signature` mirrors the reference module starkware/crypto/signature/signature.py
(same names, argument meaning and error behaviour); `starkperp.batch` adds the batch entry points
(hash / Merkle / verify / sign many) that the reference does not have.  All arithmetic runs in
libstarkperp.so on the GPU.
"""
