"""`polus` import path over the MI355X engine (`polus_amd`): user scripts written against
bioinformatics-ua/polus (`from polus.training import ClassifierTrainer`, tutorials/classifier_example.py:1-9)
run unchanged.  Every module here is a re-export; the implementation lives in polus_amd/."""
from polus_amd.context import PolusContext, logger  # noqa: F401

__version__ = "0.2.1"
