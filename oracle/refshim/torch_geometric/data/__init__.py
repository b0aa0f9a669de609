class Data(object):
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return None if self.x is None else self.x.size(0)
