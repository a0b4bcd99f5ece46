from .callbacks import ClassificationCallback, LibFMLikeCallbackBase, OrderedProbitCallback, RegressionCallback

__all__ = ["LibFMLikeCallbackBase", "RegressionCallback", "ClassificationCallback", "OrderedProbitCallback"]
