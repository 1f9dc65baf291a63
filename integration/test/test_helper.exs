ExUnit.start()
