// options.h -- command line parser for the apps.  Same surface as the
// reference's OptParser / Option<T> (inc/options.h:77-137,149-256):
// `-x value` / `--long value`, bool options are switches, require() prints
// `Required argument "<long>" not set.` and exits -1, an unknown flag or a
// missing value prints the usage text and exits 0.
#pragma once
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

class OptionBase {
public:
  OptionBase(char s, std::string l, std::string d, bool has_default)
      : _short(s), _long(std::move(l)), _desc(std::move(d)), _has_default(has_default) {}
  virtual ~OptionBase() = default;
  char getShort() const { return _short; }
  const std::string &getLong() const { return _long; }
  const std::string &getDesc() const { return _desc; }
  bool has_default() const { return _has_default; }
  virtual bool parseArgs(int argc, int &current, char **argv) = 0;
  virtual void print(std::ostream &out) const = 0;

protected:
  char _short;
  std::string _long, _desc;
  bool _has_default;
  bool _value_provided = false;
};

template <typename T> class Option : public OptionBase {
public:
  Option(char s, const std::string &l, const std::string &d) : OptionBase(s, l, d, false), _value() {}
  Option(char s, const std::string &l, const std::string &d, const T &v) : OptionBase(s, l, d, true), _value(v) {}
  T get() const { return _value; }
  T require() const {
    if (!_value_provided) {
      std::cout << "Required argument \"" << _long << "\" not set." << ENDL;
      std::exit(-1);
    }
    return _value;
  }
  operator T() const { return _value; }
  void setValue(const T &v) { _value = v; _value_provided = true; }
  bool parseArgs(int argc, int &current, char **argv) override {
    if (++current >= argc)
      return false;
    std::istringstream ss(argv[current]);
    ss >> _value;
    _value_provided = true;
    return true;
  }
  void print(std::ostream &out) const override { out << _long << ": " << _value; }

private:
  T _value;
};
template <> inline bool Option<bool>::parseArgs(int, int &, char **) {
  _value = !_value;
  _value_provided = true;
  return true;
}
template <> inline bool Option<std::string>::parseArgs(int argc, int &current, char **argv) {
  if (++current >= argc)
    return false;
  _value = argv[current]; // keep embedded spaces (paths)
  _value_provided = true;
  return true;
}

class OptParser {
public:
  explicit OptParser(std::string description) : _desc(std::move(description)) {
    _help = addOption<bool>({'h', "help", "Print help and exit.", false});
  }
  template <typename T> std::shared_ptr<Option<T>> addOption(Option<T> &&opt) {
    auto p = std::make_shared<Option<T>>(std::move(opt));
    _opts.push_back(p);
    return p;
  }
  void parse(int argc, char **argv) {
    for (int c = 1; c < argc; c++) {
      OptionBase *hit = nullptr;
      const std::string a = argv[c];
      if (a.size() >= 2 && a[0] == '-') {
        for (auto &o : _opts)
          if ((a[1] == '-' && o->getLong() == a.substr(2)) || (a[1] != '-' && o->getShort() == a[1])) {
            hit = o.get();
            break;
          }
      }
      if (!hit) {
        std::cout << "Error: Invalid argument '" << a << "'." << ENDL;
        _help->setValue(true);
        break;
      }
      if (!hit->parseArgs(argc, c, argv)) {
        std::cout << "Error: invalid argument for option " << hit->getLong() << ENDL;
        _help->setValue(true);
        break;
      }
    }
    if (_help->get())
      usage(argv[0]);
  }
  void print(std::ostream &out = std::cout) const {
    for (auto &o : _opts) {
      o->print(out);
      out << ENDL << ENDL;
    }
  }

private:
  [[noreturn]] void usage(const char *prog) const {
    std::size_t w = 0;
    for (auto &o : _opts)
      w = std::max(w, o->getLong().size());
    std::cout << "Usage:\n  " << prog << " [OPTIONS]...\nDescription:\n  " << _desc << "\nOptions:\n";
    for (auto &o : _opts) {
      if (o->getShort())
        std::cout << "  -" << o->getShort();
      else
        std::cout << "    ";
      std::cout << "  --" << std::left << std::setw((int)w) << o->getLong() << "  " << o->getDesc() << std::endl;
    }
    std::exit(0);
  }
  std::string _desc;
  std::vector<std::shared_ptr<OptionBase>> _opts;
  std::shared_ptr<Option<bool>> _help;
};
