"""Import overlay: `starkware.crypto.signature.*` with the reference's module paths, backed by
starkperp (GPU).  Put `stark-perpetual_amd/` on PYTHONPATH ahead of the reference tree to switch a
caller over without touching its imports (`starkware` itself stays a namespace package, so
`starkware.python.*` keeps resolving to the caller's own tree)."""
