"""Framework wrappers (pytensor is optional: import ``sunode_amd.wrappers.as_pytensor`` explicitly)."""
