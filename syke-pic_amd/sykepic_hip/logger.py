"""logging setup (reference ``sykepic/utils/logger.py``): ``LOGLEVEL`` env or
a YAML dictConfig named by ``LOGCONFIG``."""

import logging
import logging.config
import os


def get_logger(name):
    return logging.getLogger(name)


def setup(config_file=None):
    config_file = config_file or os.environ.get("LOGCONFIG")
    if config_file and os.path.isfile(config_file):
        import yaml
        with open(config_file) as fh:
            logging.config.dictConfig(yaml.safe_load(fh))
        return
    logging.basicConfig(level=os.environ.get("LOGLEVEL", "INFO").upper(),
                        format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
