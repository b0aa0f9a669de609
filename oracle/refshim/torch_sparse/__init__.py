class SparseTensor(object):
    """Type-only stand-in: the reference data pipeline never instantiates it."""
    pass
