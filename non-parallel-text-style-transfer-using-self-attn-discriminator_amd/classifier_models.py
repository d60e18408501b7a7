"""``getattr(models, "SelfAttnRNNClassifier")`` target mirroring reference classifier/models.py."""
from .models import ClassifierSelfAttnRNNClassifier as SelfAttnRNNClassifier  # noqa: F401
from .models import CNNClassifier  # noqa: F401,E402
