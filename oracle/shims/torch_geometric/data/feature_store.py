class TensorAttr:
    pass
