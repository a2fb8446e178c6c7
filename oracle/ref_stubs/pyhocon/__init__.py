"""Import stub so /root/reference/src/util/args.py:6 imports (never called by the hot path)."""


class ConfigFactory:
    @staticmethod
    def parse_file(path):
        raise RuntimeError("pyhocon stub: HOCON parsing is not available in this container")
