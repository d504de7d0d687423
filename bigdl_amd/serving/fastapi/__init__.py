from .api_server import FastApp  # noqa: F401
from .model_worker import ModelWorker  # noqa: F401
