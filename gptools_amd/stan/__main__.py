from . import get_include

if __name__ == "__main__":
    print(get_include())
