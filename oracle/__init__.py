"""oracle -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference hot path).

Nothing under navigation-by-deja-vu_amd/ may import this package.
"""
