class EdgeAttr:
    pass
