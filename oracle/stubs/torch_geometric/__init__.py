"""Container-only stub of torch-geometric 1.6.3 (requirement.yml:97). See oracle/stubs/README.md."""
