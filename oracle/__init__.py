"""Test infrastructure only: CPU restatements of the reference algorithms.  Never imported by cerberus_amd."""
