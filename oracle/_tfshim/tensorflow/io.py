class gfile:
    GFile = staticmethod(open)
