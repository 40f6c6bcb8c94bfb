"""python-dotenv shim (absent in the image); load_dotenv is a no-op."""


def load_dotenv(*a, **k):
    return False
